"""Resize(512) / CenterCrop(448) / ToTensor / ExpandChannels of ReportDataset.py:80-106 as demo.py:144,:251 applies
them -- the arithmetic of torchvision 0.14 (absent here, 'parity unpinned' at the library boundary): truncating Resize, half-to-even
CenterCrop offsets, /255. A synthetic PNG goes through radialog_amd.transforms and is compared with the same steps done by hand."""
import numpy as np
import pytest
import torch

from radialog_amd import transforms as T


def test_resize_truncates_and_crop_rounds_half_to_even():
    # 512 * 1000 / 601 = 851.91 -> 851 (int), where round() would give 852 (the round-1 bug)
    assert T.resized_size(601, 1000, 512) == (512, 851)
    assert T.resized_size(1000, 601, 512) == (851, 512)
    assert T.resized_size(512, 512, 512) == (512, 512) and T.resized_size(2544, 3056, 512) == (512, 615)
    # (851 - 448) / 2 = 201.5 -> 202 (half to even), floor division would give 201; (615 - 448) / 2 = 83.5 -> 84
    assert T.center_crop_box(512, 851, 448) == (32, 202, 480, 650)
    assert T.center_crop_box(615, 512, 448)[0] == 84
    assert T.center_crop_box(512, 853, 448)[1] == 202               # 202.5 -> 202: half to EVEN, not half up
    with pytest.raises(ValueError):
        T.center_crop_box(400, 512, 448)


def test_png_through_the_inference_transform(tmp_path):
    from PIL import Image
    rng = np.random.default_rng(4)
    h, w = 1000, 601
    raw = (rng.random((h, w)) * 3000 + 100).astype(np.uint16)                      # a 12-bit-ish radiograph
    raw[:50] = 0
    Image.fromarray(raw).save(tmp_path / "x.png")
    pil = T.load_image(str(tmp_path / "x.png"))
    assert pil.mode == "L" and pil.size == (w, h)
    a = np.asarray(pil)
    assert a.min() == 0 and a.max() == 255                                          # remap_to_uint8: min -> 0, max -> 255
    f = raw.astype(float)
    expect = ((f - f.min()) / (f - f.min()).max() * 255).astype(np.uint8)
    assert np.array_equal(a, expect)
    for crop in (448, 488):
        x = T.create_chest_xray_transform_for_inference(512, center_crop_size=crop)(pil)
        assert x.shape == (3, crop, crop) and x.dtype == torch.float32
        assert torch.equal(x[0], x[1]) and torch.equal(x[0], x[2])                  # ExpandChannels
        by_hand = pil.resize((512, 851), Image.BILINEAR)
        top, left = int(round((851 - crop) / 2.0)), int(round((512 - crop) / 2.0))
        by_hand = np.asarray(by_hand)[top: top + crop, left: left + crop].astype(np.float32) / 255.0
        assert np.array_equal(x[0].numpy(), by_hand)
        assert float(x.min()) >= 0.0 and float(x.max()) <= 1.0
    with pytest.raises(ValueError):
        T.ExpandChannels()(torch.zeros(3, 4, 4))


@pytest.mark.parametrize("h,w", [(1000, 601), (601, 1000), (512, 512), (615, 512), (300, 400), (1500, 1240), (520, 700), (513, 1029)])
def test_numpy_restatement_of_pillows_resample_equals_pillow_bit_for_bit(h, w):
    """oracle/pil_resize.py -- Pillow's Resample.c restated (antialiased triangle filter, C-double coefficient tables, 22-bit fixed point, two uint8 passes) plus
    torchvision's size / crop rules -- against the Pillow installed here, the library torchvision's Resize calls on a PIL image: down-scaling in both orientations,
    up-scaling (300 x 400), the skipped pass (512 wide) and no resize at all. librdx's rdx_transform_image computes the same tables in C++ and is held to the same
    bytes on the GPU (tests/test_gpu_api.py)."""
    from PIL import Image
    from oracle import pil_resize as pr
    rng = np.random.default_rng(h * 7 + w)
    a = rng.integers(0, 256, (h, w), dtype=np.uint8)
    a[: h // 7] = 0
    a[-3:, :] = 255
    nw, nh = pr.resized_size(w, h, 512)
    assert (nw, nh) == T.resized_size(w, h, 512)
    assert np.array_equal(pr.pil_resize_bilinear(a, nw, nh), np.asarray(Image.fromarray(a).resize((nw, nh), Image.BILINEAR)))
    for crop in (448, 488):
        x = T.create_chest_xray_transform_for_inference(512, center_crop_size=crop)(Image.fromarray(a)).numpy()
        assert np.array_equal(x, pr.inference_transform(a, 512, crop))
