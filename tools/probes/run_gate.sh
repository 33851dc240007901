set -x
cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_round6.py -q -m gpu -k "gated or gate_links" 2>&1 | tail -30
