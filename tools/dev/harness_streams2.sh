set -o pipefail
run() { for n in 1 3; do env $1 HARNESS_STREAMS=$n timeout -k 5 60 tools/gemm_harness $2 | sed -e "s/^/$1 /" || exit 1; done; }
run QTTS_NOP=1 "slab 64 4096 2048 1 300"
run QTTS_FULL_K_MIN_BLOCKS=64 "full_k 64 4096 2048 1 300 0"
run QTTS_FULL_K_MIN_BLOCKS=128 "full_k 64 4096 2048 1 300 0"
run QTTS_FULL_K_MIN_BLOCKS=256 "full_k 64 4096 2048 1 300 0"
run QTTS_NOP=1 "stream 64 4096 2048 1 300 0"
run QTTS_KS_GROUP=3 "k_split 64 1024 3072 0 300"
run QTTS_KS_GROUP=2 "k_split 64 1024 3072 0 300"
run QTTS_KS_GROUP=4 "k_split 64 1024 3072 0 300"
run QTTS_NOP=1 "slab 64 1024 3072 0 300"
run QTTS_NOP=1 "slab 64 1024 2048 0 300"
run QTTS_NOP=1 "k_split 64 2048 2048 1 300"
run QTTS_NOP=1 "k_split 64 2048 6144 1 300"
