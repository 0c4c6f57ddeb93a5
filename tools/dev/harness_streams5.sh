set -o pipefail
run() { for n in 1 3; do env QTTS_TUNE=alone $1 HARNESS_STREAMS=$n timeout -k 5 60 tools/gemm_harness $2 | sed -e "s/^/$1 /" || exit 1; done; }
for mb in 32 64 128 192; do
run QTTS_FULL_K_MIN_BLOCKS=$mb "full_k 64 4096 1024 0 300 0"
run QTTS_FULL_K_MIN_BLOCKS=$mb "full_k 64 6144 1024 0 300 1"
done
