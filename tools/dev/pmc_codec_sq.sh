#!/bin/bash
# Issue-side counters of the codec kernels (one pass, at most 8 SQ counters): where the waves' cycles go.
# usage: pmc_codec_sq.sh [counter ...]   sums per kernel in gpurun_out/pmc_codec_sq.sum
set -o pipefail
ctrs=${@:-SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_SALU SQ_INSTS_VALU SQ_INSTS_MFMA}
cd /tmp && export TMPDIR=/tmp REPS=1 B=${B:-16}
out=/root/repo/gpurun_out
timeout -s KILL 280 rocprofv3 --pmc $ctrs --output-format csv -d $out/pmc_codec_sq -o p -- python3 /root/repo/tools/dev/dev_codec_prof.py > $out/pmc_codec_sq.log 2>&1 || exit 1
python3 - "$out/pmc_codec_sq/p_counter_collection.csv" > $out/pmc_codec_sq.sum <<'PY'
import csv, sys, collections
tot = collections.defaultdict(collections.Counter)
for r in csv.DictReader(open(sys.argv[1])):
    k = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").replace("qtts::", "").split("(")[0][:48]
    tot[k][r["Counter_Name"]] += float(r["Counter_Value"])
names = sorted({n for c in tot.values() for n in c})
for k, c in sorted(tot.items(), key=lambda kv: -kv[1].get("SQ_WAVE_CYCLES", 0))[:14]:
    print(f"{k:48s} " + " ".join(f"{n[3:]}={c[n]:.3e}" for n in names))
PY
rm -f $out/pmc_codec_sq/p_counter_collection.csv
