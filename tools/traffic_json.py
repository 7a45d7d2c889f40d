"""Turn the raw output of tools/pmc_traffic.sh (gpurun_out/traffic_<tag>.json: FETCH_SIZE / WRITE_SIZE sums per kernel) into
the committed profile `profiles/<round>_hbm_traffic.json` that bench.py quotes as roofline.traffic:

    python tools/traffic_json.py gpurun_out/traffic_<tag>.json profiles/r02_x_hbm_traffic.json

HBM bytes per launch = 2 x FETCH_SIZE KB (gfx950 tallies a 128-byte read request as 64 B: MI355X_MICROARCH.md, HBM section; the
calibration rows of the same run check it against kernels whose byte counts are known) + WRITE_SIZE KB.  The file carries the
fingerprint of the kernel sources it was measured on; bench.py refuses to quote it for any other sources."""
import json
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))


def main():
    from bench import csrc_fingerprint

    raw = json.loads(Path(sys.argv[1]).read_text())
    out = {"source": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes, no tracing domains) over `python bench.py --steps 2 --warmup 1 "
                     "--no-cpu-baseline --no-profile` (tools/pmc_traffic.sh), averaged per dispatch of each kernel; FETCH_SIZE doubled "
                     "(gfx950 tallies 128-B read requests at 64 B: MI355X_MICROARCH.md HBM section; see `calibration`)",
           "unit": "bytes per launch", "csrc_fingerprint": csrc_fingerprint(), "kernels": {}, "calibration": {}}
    for kind, dst in (("bench", out["kernels"]), ("cal_copy", out["calibration"]), ("cal_dw", out["calibration"]), ("cal_conv3", out["calibration"])):
        fetch, write = raw.get(kind, {}).get("FETCH_SIZE", {}), raw.get(kind, {}).get("WRITE_SIZE", {})
        for k in sorted(set(fetch) | set(write), key=lambda k: -(fetch.get(k, {}).get("sum", 0) + write.get(k, {}).get("sum", 0))):
            f, w = fetch.get(k, {"avg": 0.0, "dispatches": 0}), write.get(k, {"avg": 0.0, "dispatches": 0})
            k = k.replace("s2k::", "")
            name = k if kind == "bench" else f"{kind}:{k}"
            dst[name] = {"fetch_kb_raw": round(f["avg"], 1), "write_kb": round(w["avg"], 1),
                         "hbm_bytes": round((2.0 * f["avg"] + w["avg"]) * 1024), "dispatches": max(f["dispatches"], w["dispatches"])}
    Path(sys.argv[2]).write_text(json.dumps(out, indent=1) + "\n")
    for k in ("conv_pc_kernel", "conv_igemm_kernel", "wgrad_pc_kernel", "wgrad_kernel"):
        if k in out["kernels"]:
            print(k, out["kernels"][k])


if __name__ == "__main__":
    main()
