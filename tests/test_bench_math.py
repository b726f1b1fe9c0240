"""bench.py's roofline arithmetic (algorithmic FLOPs / bytes the fractions are computed from) against independently derived figures.
CPU only: the helpers are pure functions of the configuration."""
import importlib.util
import json
import re
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench():
    spec = importlib.util.spec_from_file_location("bench_under_test", os.path.join(REPO, "bench.py"))
    mod = importlib.util.module_from_spec(spec)
    argv = sys.argv
    sys.argv = ["bench.py"]
    try:
        spec.loader.exec_module(mod)
    finally:
        sys.argv = argv
    return mod


def test_weight_stream_bytes_of_a_decode_step_match_the_parameter_count():
    from radialog_amd.config import full_cfg
    b = _bench()
    lc = full_cfg().llama
    got = b.llama_param_bytes(lc, lambda n, k: 2)
    H, I, V, L, r = 4096, 11008, 32001, 32, 8
    assert (lc.hidden, lc.inter, lc.vocab, lc.layers, lc.lora_r) == (H, I, V, L, r)
    # Vicuna-7B: q,k,v,o 4 H^2, gate/up/down 3 H I per layer, lm_head V H (the input embedding is a gather, not a stream);
    # LoRA r = 8 on q and v: A [r, H] and B [H, r] each; two RMSNorm weights per layer + the final one
    params = L * (4 * H * H + 3 * H * I + 2 * (r * H + H * r) + 2 * H) + V * H + H
    assert got == 2 * params
    assert abs(got / 1e9 - 13.22) < 0.01          # the "13.21 GB" the documents quote (13.223 GB exactly)


def test_prefill_and_encode_flops():
    from radialog_amd.config import full_cfg
    b = _bench()
    cfg = full_cfg()
    lc = cfg.llama
    H, I, V, L, r, T = lc.hidden, lc.inter, lc.vocab, lc.layers, lc.lora_r, 160
    gemm = 2.0 * T * L * ((3 * H + 2 * r) * H + H * H + 3 * H * I)                 # projections incl. the 16 LoRA-A rows
    attn = L * 2.0 * (T * T * H + T * T * H) / 2                                      # QK^T and PV, causal half
    tail = 2.0 * V * H + 2.0 * 32 * lc.qformer_dim * H                                # lm_head on the last position, img_proj on 32 slots
    assert abs(b.prefill_flops(lc, T, 1) - (gemm + attn + tail)) < 1e6
    assert abs(b.prefill_flops(lc, T, 32) - 32 * b.prefill_flops(lc, T, 1)) < 1e6
    assert abs(b.prefill_flops(lc, T, 1) / 1e12 - 2.080) < 0.002                     # the 2.08 TFLOP per prompt of the docs
    # encoder: ResNet-50 at 448 px is 4 x the 224-px 4.09 GMAC of the torchvision model card (convolutions only; no fc) = 16.4 GMAC
    v = cfg.vision
    assert (v.img, tuple(v.planes), tuple(v.blocks)) == (448, (64, 128, 256, 512), (3, 4, 6, 3))
    enc = b.encode_flops(cfg)
    assert abs(enc / 1e9 - 45.09) < 0.05                                              # 45.09 GFLOP per image (profiles / DESIGN)
    trunk_mac = 4 * 4.09e9
    assert trunk_mac * 2 < enc < trunk_mac * 2 + 14e9                                 # trunk + patch-embed conv + projector + Q-Former


def test_committed_bench_line_is_self_consistent():
    """The round's committed driver-style line: value = 1 / step time, decode fraction recomputed from its own fields."""
    name = "r04_bench.json" if os.path.exists(os.path.join(REPO, "profiles", "r04_bench.json")) else "r03_bench.json"
    with open(os.path.join(REPO, "profiles", name)) as f:
        d = json.loads(f.read().strip().splitlines()[-1])
    assert d["unit"] == "reports/s" and d["n_gpus"] == 1 and d["higher_is_better"] is True and d["vs_baseline"] is None
    assert abs(d["value"] - 1000.0 / d["ms_per_step"] * d["config"]["global_batch"]) < 1e-6
    r = d["roofline"]
    assert r["bound"] == "hbm" and r["peak"] == 8000.0 and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9
    assert abs(r["achieved"] - r["bytes_per_launch"] / r["us_per_launch"] / 1e3) < 1.0            # GB/s = bytes / us / 1e3
    assert 0.95 < r["traffic"] / r["bytes_per_launch"] < 1.05                                      # PMC traffic ~ algorithmic bytes: no re-reads
    assert d["cpu_baseline"]["kind"] == "port" and d["cpu_baseline"]["cores"] >= 1
    assert d["b32"]["value"] > d["value"] and d["fp8_b32"]["value"] > d["b32"]["value"]
    assert d["token_check"]["ok"] and d["b32"]["token_check"]["ok"] and d["fp8_b32"]["token_check"]["ok"] and d["results_verified"] is True
    assert d["cpu_baseline"]["parity"]["ok"] is True


def test_oracle_check_rule_of_the_bench_line():
    """cpu_baseline.parity: identical tokens are counted; a flip is accepted only inside 2 x the measured logit error of that step; round
    4: an ABSOLUTE logit bar per dtype on top (6e-2 fp16 / 0.45 bf16; fatal above 1.5 x), free-running or teacher-forced."""
    import torch
    b = _bench()
    V = 50
    ref = [torch.zeros(V) for _ in range(4)]
    for s, t in enumerate((7, 9, 11, 13)):
        ref[s][t] = 1.0
        ref[s][t + 1] = 0.5                  # oracle margin 0.5 everywhere
    hip = torch.stack(ref) + 0.01
    r = b.oracle_check([7, 9, 11, 13], hip, [7, 9, 11, 13], ref, "f16")
    assert r["ok"] and r["tokens_identical"] == 4 and r["divergence"] is None and abs(r["worst_logit_err"] - 0.01) < 1e-6
    assert r["abs_bar"] == 6e-2 and r["steps_over_bar"] == 0
    r = b.oracle_check([7, 9, 12, 13], hip, [7, 9, 11, 13], ref, "f16")        # flip at a margin of 0.5 with a logit error of 0.01
    assert not r["ok"] and r["tokens_identical"] == 2 and "MISMATCH" in r["divergence"]
    near = [x.clone() for x in ref]
    near[2][12] = 0.99                                                          # margin 0.01 <= 2 x 0.01
    r = b.oracle_check([7, 9, 12, 13], torch.stack(near) + 0.01, [7, 9, 11, 13], near, "f16")
    assert r["ok"] and r["tokens_identical"] == 2 and r["tokens_compared"] == 3       # free-running: the comparison ends at the flip
    r = b.oracle_check([7, 9, 12, 13], torch.stack(near) + 0.01, [7, 9, 11, 13], near, "f16", teacher_forced=True)
    assert r["ok"] and r["tokens_identical"] == 3 and r["tokens_compared"] == 4       # teacher-forced: every step is compared
    # the absolute bar: identical tokens but logits 0.1 apart -> fails in fp16 (bar 6e-2, fatal at 9e-2), passes in bf16 (0.45)
    far = torch.stack(ref) + 0.1 * torch.tensor([1.0] + [0.0] * (V - 1))
    assert not b.oracle_check([7, 9, 11, 13], far, [7, 9, 11, 13], ref, "f16")["ok"]
    rb = b.oracle_check([7, 9, 11, 13], far, [7, 9, 11, 13], ref, "bf16")
    assert rb["ok"] and rb["steps_over_bar"] == 0
    mid = torch.stack(ref) + 0.07 * torch.tensor([1.0] + [0.0] * (V - 1))      # over the bar, under 1.5 x: reported, not fatal
    rm = b.oracle_check([7, 9, 11, 13], mid, [7, 9, 11, 13], ref, "f16")
    assert rm["ok"] and rm["steps_over_bar"] == 4
    # round 5: from 16 compared steps the tests' "99 % of the steps under the bar" is ENFORCED -- over 32 steps one may sit
    # between the bar and 1.5 x, two may not
    toks32 = [7] * 32
    ref32 = [ref[0].clone() for _ in range(32)]
    bump = lambda n: torch.stack(ref32) + 0.07 * torch.tensor([[1.0] + [0.0] * (V - 1)] * n + [[0.0] * V] * (32 - n))      # noqa: E731
    r1 = b.oracle_check(toks32, bump(1), toks32, ref32, "f16", teacher_forced=True)
    assert r1["ok"] and r1["steps_over_bar"] == 1 and r1["steps_over_bar_allowed"] == 1
    r2 = b.oracle_check(toks32, bump(2), toks32, ref32, "f16", teacher_forced=True)
    assert not r2["ok"] and r2["steps_over_bar"] == 2 and "over the absolute bar" in r2["divergence"]
    # an explicit bar (the fp8 leg: 0.5, the end-to-end fp16 leg: 0.25) replaces the dtype's
    assert b.oracle_check(toks32, bump(2), toks32, ref32, "f16", teacher_forced=True, bar=0.25)["ok"]


def test_launcher_line_gpus_resolution_and_multi_rank_baseline_pointer():
    """Ready for the 8-GPU node: `python bench.py --gpus 8` becomes the driver's own launcher line; --gpus defaults to
    the launcher's WORLD_SIZE; an explicit conflict exits 2; N > 1 lines carry a cpu_baseline pointer to the N = 1 figure."""
    import types
    import pytest
    b = _bench()
    cmd = b.spawn_command(8, 29511, ["--gpus", "8", "--steps", "3", "--warmup", "1"])
    assert cmd[1:4] == ["-m", "torch.distributed.run", "--nnodes=1"] and "--nproc-per-node=8" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and cmd[cmd.index("--master-port") + 1] == "29511"
    assert cmd[-7].endswith("bench.py") and cmd[-6:] == ["--gpus", "8", "--steps", "3", "--warmup", "1"]
    a = types.SimpleNamespace(gpus=None)
    b.resolve_gpus(a, {})
    assert a.gpus == 1
    a = types.SimpleNamespace(gpus=None)
    b.resolve_gpus(a, {"WORLD_SIZE": "8", "RANK": "3"})
    assert a.gpus == 8
    a = types.SimpleNamespace(gpus=8)
    b.resolve_gpus(a, {"WORLD_SIZE": "8", "RANK": "0"})
    assert a.gpus == 8
    with pytest.raises(SystemExit) as e:
        b.resolve_gpus(types.SimpleNamespace(gpus=2), {"WORLD_SIZE": "4", "RANK": "0"})
    assert e.value.code == 2
    ptr = b.cpu_baseline_pointer()
    assert ptr["kind"] == "port" and ptr["unit"] == "reports/s" and ptr["value"] and ptr["source"].startswith("profiles/")


def test_fixture_check_compares_row0_tokens_with_the_committed_file():
    b = _bench()
    with open(os.path.join(REPO, "tests", "golden", "bench_tokens.json")) as f:
        fx = json.load(f)
    key = "bf16_b1_row0"
    assert key in fx and len(fx[key]) == 8
    assert b.fixture_check("bf16", 1, False, list(fx[key]))["ok"]
    bad = list(fx[key]); bad[3] += 1
    assert not b.fixture_check("bf16", 1, False, bad)["ok"]
    assert b.fixture_check("f16", 7, False, [1] * 8)["expected"] is None       # unprofiled configuration: reported, not failed


def test_gpus_n_without_a_launcher_fails_loudly_when_the_gpus_are_not_there():
    """`python bench.py --gpus N` must start N ranks itself or refuse -- never benchmark one GPU and print n_gpus 1."""
    import subprocess
    import torch
    if torch.cuda.device_count() >= 2:
        return
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    p = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "2"], env=env, capture_output=True, text=True, timeout=300)
    assert p.returncode == 2 and p.stdout.strip() == "" and "only" in p.stderr and "GPU" in p.stderr
    env.update(RANK="0", WORLD_SIZE="4", LOCAL_RANK="0")                          # a launcher with the wrong rank count
    p = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "2"], env=env, capture_output=True, text=True, timeout=300)
    assert p.returncode == 2 and p.stdout.strip() == "" and "WORLD_SIZE" in p.stderr


def test_committed_round6_line_is_compact_and_agrees_with_its_detail_record():
    """Round 6: the stdout line is a < 4 KB extract of the full record (bench.compact_line); both are committed (profiles/r06_bench.json = the line the run printed,
    r06_bench_detail.json = the record beside it). The extract must be reproducible from the record, self-consistent, and carry the BASELINE configs[2] / [4] numbers
    as scalars -- top level AND inside `roofline`, whose scalars the driver's record keeps."""
    b = _bench()
    with open(os.path.join(REPO, "profiles", "r06_bench.json")) as f:
        raw = f.read().strip().splitlines()[-1]
    d = json.loads(raw)
    with open(os.path.join(REPO, "profiles", "r06_bench_detail.json")) as f:
        full = json.load(f)
    assert len(raw) < 4096
    assert d == json.loads(json.dumps(b.compact_line(full, d["detail"])))
    assert d["unit"] == "reports/s" and d["n_gpus"] == 1 and d["higher_is_better"] is True and d["vs_baseline"] is None and d["dtype"] == "bf16"
    assert abs(d["value"] - 1000.0 / d["ms_per_step"] * d["config"]["global_batch"]) < 1e-6
    r = d["roofline"]
    assert r["bound"] == "hbm" and r["peak"] == 8000.0 and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-4
    assert abs(r["achieved"] - r["bytes_per_launch"] / r["us_per_launch"] / 1e3) < 1.0 and 0.95 < r["traffic"] / r["bytes_per_launch"] < 1.05
    for key in ("b32", "fp8_b32", "b64", "b128", "fp8_b128"):
        assert d[f"{key}_value"] > d["value"] and r[f"{key}_decode_avg_frac"] == d[f"{key}_frac"]
        assert abs(d[f"{key}_value"] / full[key]["value"] - 1) < 1e-4
    assert d["b32_global_batch"] == 32 and d["fp8_b32_value"] > d["b32_value"]
    assert d["cpu_baseline"]["kind"] == "port" and d["cpu_baseline"]["cores"] >= 1 and d["cpu_baseline"]["value"] > 0
    assert set(d["parity"]) >= {"bf16", "f16", "b32", "fp8", "e2e_f16"} and all(v["ok"] for v in d["parity"].values())
    assert d["results_verified"] is True and d["oracle_checked"] is True and re.fullmatch(r"[0-9a-f]{16}", d["build_hash"])
