set -o pipefail
run() { for n in 1 3; do env $1 HARNESS_STREAMS=$n timeout -k 5 60 tools/gemm_harness $2 | sed -e "s/^/$1 /" || exit 1; done; }
for g in 3 6; do
run "QTTS_KS_GROUP=$g" "k_split 64 1024 2048 0 300"
run "QTTS_KS_GROUP=$g" "k_split 64 1024 3072 0 300"
done
