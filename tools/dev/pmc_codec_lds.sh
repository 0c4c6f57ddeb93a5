#!/bin/bash
# LDS bank conflicts of the codec kernels: SQ_LDS_BANK_CONFLICT (extra cycles) against SQ_LDS_IDX_ACTIVE (all LDS-array cycles) and
# SQ_BUSY_CYCLES, one pass over tools/dev/dev_codec_prof.py with one repetition; per-kernel sums in gpurun_out/pmc_codec_lds.sum.
set -o pipefail
cd /tmp && export TMPDIR=/tmp REPS=1 B=${B:-16}
out=/root/repo/gpurun_out
timeout -s KILL 280 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAVE_CYCLES --output-format csv -d $out/pmc_codec_lds -o p -- python3 /root/repo/tools/dev/dev_codec_prof.py > $out/pmc_codec_lds.log 2>&1 || exit 1
python3 - "$out/pmc_codec_lds/p_counter_collection.csv" > $out/pmc_codec_lds.sum <<'PY'
import csv, sys, collections
tot = collections.defaultdict(collections.Counter)
for r in csv.DictReader(open(sys.argv[1])):
    k = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").replace("qtts::", "").split("(")[0][:60]
    tot[k][r["Counter_Name"]] += float(r["Counter_Value"])
for k, c in sorted(tot.items(), key=lambda kv: -kv[1]["SQ_LDS_IDX_ACTIVE"]):
    a = c["SQ_LDS_IDX_ACTIVE"]
    print(f"{k:60s} lds_active {a:.3e} conflict {c['SQ_LDS_BANK_CONFLICT']:.3e} ({c['SQ_LDS_BANK_CONFLICT'] / a if a else 0:.2f}) insts {c['SQ_INSTS_LDS']:.3e} wave_cycles {c['SQ_WAVE_CYCLES']:.3e}")
PY
rm -f $out/pmc_codec_lds/p_counter_collection.csv
