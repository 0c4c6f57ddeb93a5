#!/bin/bash
# HBM-side traffic of one codec decode (B x 375 frames, bf16, reference-default shapes): FETCH_SIZE and WRITE_SIZE in separate
# passes over tools/dev/dev_codec_prof.py with one repetition; per-kernel sums in KB land in gpurun_out/pmc_codec_<counter>.sum.
set -o pipefail
cd /tmp && export TMPDIR=/tmp REPS=1 B=${B:-16}
out=/root/repo/gpurun_out
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -s KILL 280 rocprofv3 --pmc $c --output-format csv -d $out/pmc_codec_$c -o p -- python3 /root/repo/tools/dev/dev_codec_prof.py > $out/pmc_codec_$c.log 2>&1 || exit 1
  python3 - "$out/pmc_codec_$c/p_counter_collection.csv" "$c" > $out/pmc_codec_$c.sum <<'PY'
import csv, sys, collections
tot = collections.Counter(); n = collections.Counter()
for r in csv.DictReader(open(sys.argv[1])):
    if r["Counter_Name"] == sys.argv[2]:
        k = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").replace("qtts::", "").split("(")[0][:70]
        tot[k] += float(r["Counter_Value"]); n[k] += 1
for k, v in tot.most_common():
    print(f"{v:.1f}\t{n[k]}\t{k}")
PY
  rm -f $out/pmc_codec_$c/p_counter_collection.csv
done
