export SS_LIB_PATH=$PWD/spaghettisearch_amd/libspaghetti_rank_kmask.so
for m in 0xFFFFFFFF 0x1 0x20 0x10 0x8 0x21 0x31; do
  SS_PR_KIND_MASK=$m R=3 timeout -k 10 200 python tools/pr_exp.py 2>&1 | grep -E "lib=|\[pr\]" | sed "s/^/mask $m: /" >> gpurun_out/r4a_kmask.log || exit 1
done
