set -o pipefail
run() { for n in 1 3; do env $1 HARNESS_STREAMS=$n timeout -k 5 60 tools/gemm_harness $2 | sed -e "s/^/$1 /" || exit 1; done; }
run QTTS_NOP=1 "stream 64 4096 1024 0 300 0"
run QTTS_NOP=1 "stream 64 6144 1024 0 300 1"
run QTTS_NOP=1 "stream 64 2048 1024 0 300 0"
