#!/bin/bash
# MFMA-busy evidence for the headline step (run on the GPU box): tools/pmc_mfma_busy.sh <tag> [extra bench.py args]
# One --pmc pass (SQ counters + GRBM_GUI_ACTIVE, no tracing domains) over `python3 bench.py --steps 2 --warmup 1`; writes
# gpurun_out/<tag>_pmc_mfma_busy.md: per MFMA kernel (all dispatches summed) the matrix-core busy cycles against the cycles the
# 1,024 SIMDs were available (GRBM_GUI_ACTIVE is summed over the 8 XCDs: / 8 = shader cycles of the dispatch).
tag=$1; shift
root=$(pwd)
out=$root/gpurun_out/pmcbusy_$tag
rm -rf $out; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_INSTS_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE \
  -d $out/p -o p --output-format csv -- python3 $root/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-profile --no-prithvi "$@" > $out/run.log 2>&1 || tail -5 $out/run.log
python3 - <<PY
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for f in glob.glob("$out/p/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0].split("<")[0].replace("void ", "").replace("s2k::", "")
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
        if r["Counter_Name"] == "SQ_INSTS_VALU": n[k] += 1
want = ["conv_pc_kernel", "wgrad_pc_kernel", "conv_igemm_kernel", "wgrad_kernel", "conv_dma_kernel", "conv_q4_kernel", "conv_bf16_kernel", "wgrad_bf16_kernel", "attn_fwd_lds_kernel"]
lines = ["# $tag: matrix-core busy counters of the MFMA kernels, headline step ($*)", "",
         "source: rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_INSTS_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE",
         "-- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-profile --no-prithvi $*  (tools/pmc_mfma_busy.sh; all dispatches of a kernel summed)", "",
         "MFMA busy % = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x GRBM_GUI_ACTIVE / 8 XCDs); f32 32x32x2 = 64 busy cycles per instruction, bf16 32x32x16 = 32.",
         "Profiled passes run at a lower clock than un-profiled ones (MI355X_MICROARCH.md, DVFS): ratios, not absolute times.", "",
         "| kernel | dispatches | MFMA busy % of all SIMD cycles | MFMA insts (M) | busy cycles / MFMA inst | VALU insts per MFMA inst | wave cycles: issue-stalled % | active % |",
         "|---|---:|---:|---:|---:|---:|---:|---:|"]
for k in want + sorted(set(acc) - set(want), key=lambda k: -acc[k]["SQ_VALU_MFMA_BUSY_CYCLES"])[:6]:
    d = acc.get(k)
    if not d or not d["SQ_INSTS_MFMA"]: continue
    cyc = d["GRBM_GUI_ACTIVE"] / 8.0
    wc = d["SQ_WAVE_CYCLES"] or 1.0
    lines.append("| \`%s\` | %d | %.1f | %.2f | %.1f | %.2f | %.1f | %.1f |" % (
        k, n[k], 100.0 * d["SQ_VALU_MFMA_BUSY_CYCLES"] / (1024.0 * cyc) if cyc else 0.0, d["SQ_INSTS_MFMA"] / 1e6,
        d["SQ_VALU_MFMA_BUSY_CYCLES"] / d["SQ_INSTS_MFMA"], d["SQ_INSTS_VALU"] / d["SQ_INSTS_MFMA"],
        100.0 * d["SQ_WAIT_INST_ANY"] / wc, 100.0 * d["SQ_ACTIVE_INST_ANY"] / wc))
open("$root/gpurun_out/${tag}_pmc_mfma_busy.md", "w").write("\n".join(lines) + "\n")
print("\n".join(lines))
PY
