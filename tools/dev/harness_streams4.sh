set -o pipefail
run() { for n in 3; do env $1 HARNESS_STREAMS=$n timeout -k 5 60 tools/gemm_harness $2 | sed -e "s/^/$1 /" || exit 1; done; }
for mb in 64 96 128 192 256 384; do
run QTTS_GEMM_MIN_BLOCKS=$mb "slab 64 2048 6144 1 300"
run QTTS_GEMM_MIN_BLOCKS=$mb "slab 64 2048 2048 1 300"
done
run "QTTS_GEMM_NCH=4" "slab 64 2048 6144 1 300"
run "QTTS_GEMM_NCH=16" "slab 64 2048 6144 1 300"
run "QTTS_GEMM_NT=1" "slab 64 2048 6144 1 300"
run "QTTS_GEMM_KIND=0" "slab 64 2048 6144 1 300"
run "QTTS_GEMM_KIND=0" "slab 64 2048 2048 1 300"
