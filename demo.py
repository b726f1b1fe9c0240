#!/usr/bin/env python
"""demo.py -- the reference's interactive entry point (demo.py:245-305 `get_response`) on the MI355X-native path.

Same flow and flags (`--cfg-path`, `--options`): image -> vis_transforms (Resize 512 / CenterCrop 448 / ToTensor /
ExpandChannels, demo.py:144) -> blip_model.forward_image -> report prompt with 32 x <IMG> -> lang_model.generate
(greedy, max_new_tokens=300) -> text after "ASSISTANT:". The gradio widgets of the reference are UI only and are not
reproduced; `--image` runs one request from the command line (a synthetic radiograph when no path is given, random-init
weights when no checkpoints are reachable -- there is no network here)."""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from radialog_amd import synth                                              # noqa: E402
from radialog_amd.blip2_qformer import Config, tasks                        # noqa: E402
from radialog_amd.chexpert_model import CHEXPERT_COLS, ChexpertClassifier   # noqa: E402
from radialog_amd.modeling_llama_imgemb import LlamaForCausalLM, PeftModelForCausalLM   # noqa: E402
from radialog_amd.prompter import new_conversation, report_prompt          # noqa: E402
from radialog_amd.tokenizer import load_tokenizer                          # noqa: E402


def parse_args(argv=None):
    p = argparse.ArgumentParser(description="RaDialog demo (MI355X-native hot path)")
    p.add_argument("--cfg-path", default=os.path.join(os.path.dirname(os.path.abspath(__file__)), "configs", "blip2_pretrain_stage1_emb.yaml"))
    p.add_argument("--options", nargs="+")
    p.add_argument("--image", default=None, help=".png/.jpg chest X-ray; default: synthetic 448x448 image")
    p.add_argument("--findings", default=None, help="predicted-findings text for the prompt; default: run the findings classifier "
                   "(demo.py:256-261), or 'no finding' with --no-classifier")
    p.add_argument("--no-classifier", action="store_true")
    p.add_argument("--chexpert_ckpt", default=None, help="ChexpertClassifier Lightning checkpoint (random-init when absent)")
    p.add_argument("--vicuna", default=None, help="local lmsys/vicuna-7b-v1.3 directory (weights + tokenizer)")
    p.add_argument("--lora_model", default=None, help="adapter dir (adapter_model.bin incl. img_proj_layer)")
    p.add_argument("--max_new_tokens", type=int, default=300)
    p.add_argument("--dtype", default="bf16", choices=["bf16", "f16"])
    p.add_argument("--fp8", action="store_true", help="BASELINE configs[4]: decoder GEMMs in OCP e4m3 on the fp8 MFMA (weights_fp8=True)")
    p.add_argument("--synthetic", action="store_true", help="run on the deterministic random-init weights (no checkpoints are "
                   "reachable without a network); without it every missing weight file is an error")
    args = p.parse_args(argv)
    if args.synthetic:
        args.options = (args.options or []) + ["model.synthetic=true"]
    return args


def load_image(path, crop=448, engine=None):
    """demo.py:205-218 (load_image / remap_to_uint8 -> PIL 'L') + the inference transform (ReportDataset.py:96-106, demo.py:144);
    crop=488 gives the findings classifier's `cp_transforms` (demo.py:169). With an engine the resize / crop / ToTensor run on its GPU (rdx_transform_image:
    the same bytes as the PIL path, tests/test_gpu_api.py)."""
    from radialog_amd import transforms
    return transforms.create_chest_xray_transform_for_inference(512, center_crop_size=crop, engine=engine)(transforms.load_image(path))


def init_blip(cfg):
    task = tasks.setup_task(cfg)
    return task.build_model(cfg).to(torch.device("cpu"))


def init_vicuna(args):
    tok = load_tokenizer(args.vicuna)
    dt = torch.float16 if args.dtype == "f16" else torch.bfloat16
    lang_model = LlamaForCausalLM.from_pretrained(args.vicuna, torch_dtype=dt, device_map="auto", max_batch=1, max_len=1024,
                                                  synthetic=args.synthetic, weights_fp8=args.fp8)
    if args.lora_model:
        lang_model = PeftModelForCausalLM.from_pretrained(lang_model, args.lora_model, torch_dtype=dt)
    lang_model.reuse_prefix_kv = True       # chat turns re-send the whole conversation: keep the KV rows of the shared token prefix
    return lang_model.eval(), tok


def init_chexpert_predictor(args):
    """demo.py:155-170: the classifier that fills "Predicted Findings:" for images without precomputed labels."""
    if args.chexpert_ckpt:
        return ChexpertClassifier.load_from_checkpoint(args.chexpert_ckpt, num_classes=14, class_names=CHEXPERT_COLS).eval().half()
    if not args.synthetic:
        raise RuntimeError("no --chexpert_ckpt given (use --findings / --no-classifier, or --synthetic for random-init weights)")
    from radialog_amd.engine import synth_getter
    m = ChexpertClassifier(num_classes=14)
    return m.set_weight_getter(synth_getter(m.cfg, torch.device("cuda", 0), lora=False)).eval().half()


def get_response(blip_model, lang_model, tok, conv, image, findings, max_new_tokens=300):
    blip_model = blip_model.to(torch.device("cuda"))
    qformer_embs = blip_model.forward_image(image[None].to(torch.device("cuda")))[0].cpu().detach()
    blip_model = blip_model.to(torch.device("cpu"))
    torch.save(qformer_embs, "current_chat_img.pt")                           # the reference's hand-off file (demo.py:273)
    conv.append_message(conv.roles[0], report_prompt(findings))
    conv.append_message(conv.roles[1], None)
    inputs = tok(conv.get_prompt(), return_tensors="pt")
    out = lang_model.generate(input_ids=inputs["input_ids"], dicom=None, use_img=True, return_dict_in_generate=True,
                              output_scores=True, max_new_tokens=max_new_tokens)
    preds = tok.batch_decode(out.sequences, skip_special_tokens=True)
    new_pred = preds[0].split("ASSISTANT:")[-1]
    conv.messages.pop()
    conv.append_message(conv.roles[1], new_pred)
    return new_pred, out


def main(argv=None):
    """Runs one request; returns {"findings", "prediction", "sequences", "n_scores"} (the reference's script returns nothing; the
    return value exists so that tests/test_gpu_entrypoints.py can assert on what was printed)."""
    args = parse_args(argv)
    cfg = Config(args)
    blip_model = init_blip(cfg).eval()
    lang_model, tok = init_vicuna(args)
    tf_engine = None
    if args.image:                                      # the image transform runs on the decoder context's GPU (it needs no weights)
        lang_model._ensure_engine()
        tf_engine = lang_model._engine
    image = load_image(args.image, engine=tf_engine) if args.image else synth.synth_images(1, 448)[0]
    findings = args.findings
    if findings is None and not args.no_classifier:
        cp_image = load_image(args.image, crop=488, engine=tf_engine) if args.image else synth.synth_images(1, 488)[0]
        findings = init_chexpert_predictor(args).predict_findings(cp_image[None].half().cuda())[0]
        print("predicted findings:", findings or "(none)")
    findings = findings or "no finding"
    pred, out = get_response(blip_model, lang_model, tok, new_conversation(), image, findings, args.max_new_tokens)
    print(f"generated {out.sequences.shape[1]} ids, {len(out.scores)} steps")
    print("ASSISTANT:", pred[:400])
    lang_model.close()
    return {"findings": findings, "prediction": pred, "sequences": out.sequences.cpu(), "n_scores": len(out.scores)}


if __name__ == "__main__":
    main()
