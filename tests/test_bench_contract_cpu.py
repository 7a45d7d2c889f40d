"""bench.py's host-side logic (no GPU): the roofline object, the traffic provenance rule and the CPU description."""
import json
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
import bench  # noqa: E402


def _kernels():
    return {"conv_pc_kernel": {"ms": 10.0, "launches": 50, "flops": 1.0e12, "bytes": 5.0e9},
            "wgrad_pc_kernel": {"ms": 6.0, "launches": 40, "flops": 0.5e12, "bytes": 4.0e9},
            "conv_igemm_kernel": {"ms": 8.0, "launches": 100, "flops": 0.35e12, "bytes": 7.0e9}}


def test_roofline_names_the_kernel_with_the_most_time_and_prices_it_against_spec_and_measured_peak():
    r = bench.make_roofline(_kernels(), {"mfma_f32_tflops": 150.0, "mfma_clock_mhz": 2300, "copy_gbps": 4600.0}, traffic_workload=False)
    assert r["kernel"] == "conv_pc_kernel" and r["bound"] == "mfma" and r["unit"] == "TFLOP/s"
    assert abs(r["achieved"] - 100.0) < 1e-9 and abs(r["frac"] - round(100.0 / bench.PEAK_F32_MFMA_TFLOPS, 4)) < 1e-9
    assert r["peak"] == bench.PEAK_F32_MFMA_TFLOPS and r["peak_measured"] == 150.0 and abs(r["frac_of_measured"] - round(100.0 / 150.0, 4)) < 1e-9
    assert r["avg_launch_ms"] == 0.2 and r["launches"] == 50 and r["algorithmic_bytes_per_launch"] == 100_000_000
    assert r["traffic"] is None and "U-Net" in r["traffic_source"]
    assert list(r["mfma_kernels"]) == ["conv_pc_kernel", "conv_igemm_kernel", "wgrad_pc_kernel"]      # by device time
    assert abs(r["all_mfma_tflops"] - round(1.85e12 / 24e-3 / 1e12, 2)) < 1e-9


def test_traffic_is_quoted_only_for_the_kernel_sources_it_was_measured_on(tmp_path, monkeypatch, capsys):
    prof = tmp_path / "profiles"
    prof.mkdir()
    monkeypatch.setattr(bench, "ROOT", tmp_path)
    monkeypatch.setattr(bench, "csrc_fingerprint", lambda: "aaaa")
    assert bench.measured_traffic("conv_pc_kernel")[0] is None                      # no file at all
    doc = {"csrc_fingerprint": "bbbb", "kernels": {"conv_pc_kernel": {"hbm_bytes": 123}}}
    (prof / "r09_hbm_traffic.json").write_text(json.dumps(doc))
    val, src = bench.measured_traffic("conv_pc_kernel")
    assert val is None and src.startswith("stale") and "re-run tools/pmc_traffic.sh" in capsys.readouterr().err   # loud
    doc["csrc_fingerprint"] = "aaaa"
    (prof / "r09_hbm_traffic.json").write_text(json.dumps(doc))
    val, src = bench.measured_traffic("conv_pc_kernel")
    assert val == 123 and "not measured in this run" in src
    assert bench.measured_traffic("no_such_kernel")[0] is None


def test_committed_traffic_profile_matches_the_committed_kernel_sources():
    """the file the default bench line quotes must have been measured on the sources in this tree"""
    files = sorted((ROOT / "profiles").glob("*_hbm_traffic.json"))
    assert files, "no committed PMC traffic profile"
    import pytest

    doc = json.loads(files[-1].read_text())
    if doc["csrc_fingerprint"] != bench.csrc_fingerprint():
        # bench.py then reports traffic = null and says so on stderr; a kernel edit in progress must not break the CPU suite
        pytest.skip(f"{files[-1].name} is stale (kernel sources changed since the PMC run): re-run tools/pmc_traffic.sh")
    for k in ("conv_pc_kernel", "wgrad_pc_kernel", "conv_igemm_kernel", "wgrad_kernel"):
        assert doc["kernels"][k]["hbm_bytes"] > 0


def test_cpu_description_and_stage_work():
    c = bench.cpu_model()
    assert c["logical_cpus"] >= 1 and c["usable_cpus"] >= 1 and "cgroup_cpu_quota" in c
    import numpy as np

    import s2lc_amd  # noqa: F401
    from s2lc_amd.plan import opdefs as D
    from s2lc_amd.plan.program import Arena, Program

    ar = Arena(D.BASE["WS"])
    P, Q, W = ar.alloc("p", (2, 8, 4, 4)), ar.alloc("q", (2, 16, 4, 4)), ar.alloc("w", (1, 8, 16))
    prog = Program()
    prog.add("WGRAD", P=P, BNVP=None, GATEP=None, Q=Q, BNVQ=None, GATEQ=None, WGS=W, B=2, M=8, C=16, CTOT=16, H=4, W=4, KH=1, KW=1,
             STRIDE=1, PAD_T=0, PAD_L=0, HO=4, WO=4, PROP=0, PROQ=0, MODE=0)
    kind, fl, by = bench.stage_work(prog.pack()[0], D)
    assert kind == "WGRAD" and fl == 2.0 * 8 * 16 * 32 and by == 4.0 * (2 * 8 * 16 + 2 * 16 * 16 + 8 * 16)
    assert isinstance(np.asarray(prog.pack()), np.ndarray)


def test_launcher_starts_one_process_per_rank_and_returns_the_worst_exit_code(tmp_path, capfd):
    """`python bench.py --gpus N` (no rank environment) becomes a launcher: N children with RANK / LOCAL_RANK / WORLD_SIZE /
    MASTER_ADDR=127.0.0.1 / MASTER_PORT, only rank 0 on stdout, worst exit code returned (stand-in rank script: no GPU here)."""
    script = tmp_path / "rank.py"
    script.write_text("import os, sys\n"
                      "r = int(os.environ['RANK'])\n"
                      "assert os.environ['WORLD_SIZE'] == '3' and os.environ['LOCAL_RANK'] == str(r)\n"
                      "assert os.environ['MASTER_ADDR'] == '127.0.0.1' and int(os.environ['MASTER_PORT']) > 0\n"
                      "assert os.environ['HSA_ENABLE_IPC_MODE_LEGACY'] == '0'\n"
                      "print('[Gloo] chatter from rank', r)\n"
                      "print('{\"metric\": \"m\", \"rank\": %d}' % r)\n"
                      "sys.exit(int(sys.argv[1]) if r == 2 else 0)\n")
    assert bench.launch_ranks(3, ["0"], script=str(script)) == 0
    cap = capfd.readouterr()
    lines = [ln for ln in cap.out.splitlines() if ln.strip()]
    assert lines == ['{"metric": "m", "rank": 0}']              # rank 0's result line only: ONE JSON line on stdout
    assert "[Gloo] chatter from rank 0" in cap.err and "rank 1" not in cap.err
    assert bench.launch_ranks(3, ["7"], script=str(script)) == 7


def test_launcher_ends_the_other_ranks_when_one_dies_early(tmp_path, capfd):
    """a rank k > 0 that dies while rank 0 sits in a collective (here: sleeps with its stdout open) must not hang the
    launcher until rank 0's own distributed timeout - the launcher polls every rank, ends the survivors and returns the code."""
    import time

    script = tmp_path / "rank.py"
    script.write_text("import os, sys, time\n"
                      "r = int(os.environ['RANK'])\n"
                      "if r == 1:\n"
                      "    sys.exit(9)\n"                      # dies at once (import error, out of memory, RCCL init failure ...)
                      "print('rank', r, 'waiting in a collective', flush=True)\n"
                      "time.sleep(120)\n")
    t0 = time.monotonic()
    assert bench.launch_ranks(3, [], script=str(script)) == 9
    assert time.monotonic() - t0 < 30.0
    assert "rank 1 exited with code 9" in capfd.readouterr().err


def test_launcher_deadline(tmp_path, capfd, monkeypatch):
    """every rank hangs: the overall deadline ends the run with a non-zero code"""
    script = tmp_path / "rank.py"
    script.write_text("import time\ntime.sleep(120)\n")
    monkeypatch.setenv("S2K_LAUNCH_DEADLINE_S", "2")
    assert bench.launch_ranks(2, [], script=str(script)) == 124
    assert "launch deadline" in capfd.readouterr().err
