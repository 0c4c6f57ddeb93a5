#!/bin/bash
# LDS bank conflicts and issue-side counters of the decode kernels: one eager (graph-less) single-stream run of 12 frames.
# usage: pmc_frame_lds.sh alone|shared     sums per kernel in gpurun_out/pmc_frame_lds_<tag>.sum
set -o pipefail
tag=${1:-alone}
export QTTS_TUNE=$tag
cd /tmp && export TMPDIR=/tmp
out=/root/repo/gpurun_out
timeout -s KILL 280 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_MFMA --output-format csv -d $out/pmc_frame_lds_$tag -o p -- python3 /root/repo/bench.py --no-graph --steps 1 --warmup 0 --pipeline 1 --no-codec --no-cpu-baseline --frames 12 > $out/pmc_frame_lds_$tag.log 2>&1 || exit 1
python3 - "$out/pmc_frame_lds_$tag/p_counter_collection.csv" > $out/pmc_frame_lds_$tag.sum <<'PY'
import csv, sys, collections
tot = collections.defaultdict(collections.Counter); n = collections.Counter()
for r in csv.DictReader(open(sys.argv[1])):
    k = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").replace("qtts::", "").split("(")[0][:46]
    tot[k][r["Counter_Name"][3:]] += float(r["Counter_Value"])
    if r["Counter_Name"] == "SQ_WAVE_CYCLES": n[k] += 1
for k, c in sorted(tot.items(), key=lambda kv: -kv[1]["WAVE_CYCLES"])[:22]:
    a = c["LDS_IDX_ACTIVE"]
    print(f"{k:46s} n={n[k]:5d} wave_cyc {c['WAVE_CYCLES']:.2e} wait {c['WAIT_ANY'] / max(c['WAVE_CYCLES'], 1):.2f} lds_active {a:.2e} conflict {c['LDS_BANK_CONFLICT'] / a if a else 0:.2f} valu {c['INSTS_VALU']:.2e} salu {c['INSTS_SALU']:.2e} mfma {c['INSTS_MFMA']:.2e} lds {c['INSTS_LDS']:.2e}")
PY
rm -f $out/pmc_frame_lds_$tag/p_counter_collection.csv
