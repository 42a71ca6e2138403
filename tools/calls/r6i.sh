cd "${GRAFT_REPO_ROOT}" || exit 1
export TMPDIR=/tmp
for w in 1 0; do for s in 0 2; do
echo "== SSA_GEMM_WIDE=$w SMALL=$s"; SSA_GEMM_WIDE=$w SSA_GEMM_WIDE_SMALL=$s timeout 300 python -m pytest tests/test_e2e_gpu.py -q -x -m gpu -k test_train_step -s 2>&1 | grep -E "running stats|passed|failed|grad cosine|train loss"
done; done
echo "== WGRAD_ALL=0"; SSA_WGRAD_ALL=0 SSA_GEMM_WIDE=0 timeout 300 python -m pytest tests/test_e2e_gpu.py -q -x -m gpu -k test_train_step -s 2>&1 | grep -E "running stats|passed|failed|grad cosine|train loss"
