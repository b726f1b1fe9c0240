#!/usr/bin/env python
"""Regenerate tests/golden/bench_tokens.json on an MI355X: the first 8 greedy tokens of row 0 of every configuration bench.py times
(batch 1, 32, 64, 128; fp8 weights at batch 32 and 128; fp16 at batch 1), taken from the `token_check` objects of one bench line. Run after any kernel change that
re-orders an accumulation (`python tools/make_bench_fixture.py`, ~1.5 min); the batch-1 line is additionally checked against the
full-depth CPU oracle by every default bench run (cpu_baseline.parity), so a wrong fixture cannot hide a wrong kernel."""
import json
import os
import subprocess
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    out = {"_note": "first 8 greedy tokens of row 0 of bench.py's timed configurations, written by tools/make_bench_fixture.py on an MI355X"}
    for extra in ([], ["--dtype", "f16", "--no-fp8"]):
        detail = os.path.join("/tmp", f"rdx_fixture_detail_{os.getpid()}.json")        # round 6: the nested records are in the detail file, the stdout line is their extract
        p = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--steps", "1", "--warmup", "0", "--no-cpu-baseline"] + extra,
                           capture_output=True, text=True, env=dict(os.environ, RDX_BENCH_DETAIL=detail))
        if not os.path.exists(detail) or not [l for l in p.stdout.splitlines() if l.startswith("{")]:
            sys.stderr.write(p.stderr[-4000:])
            raise SystemExit("bench.py printed no JSON line")
        with open(detail) as f:
            d = json.load(f)
        os.remove(detail)
        for obj in [d] + [d[k] for k in ("b32", "fp8_b32", "b64", "b128", "fp8_b128", "f16_b1") if k in d]:
            tc = obj["token_check"]
            out[tc["key"]] = tc["tokens"]
    path = os.path.join(REPO, "tests", "golden", "bench_tokens.json")
    with open(path, "w") as f:
        json.dump(out, f, indent=1)
        f.write("\n")
    print(json.dumps(out))


if __name__ == "__main__":
    main()
