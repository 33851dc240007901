cd $GRAFT_REPO_ROOT
for v in 2 1 0 2 1; do echo "== CHEBGCN_SMALL_LAUNCH_X=$v"; CHEBGCN_SMALL_LAUNCH_X=$v python tools/refshape.py --nodes 1000 | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['eager']['ms_per_step'], d['hip_graph']['ms_per_step'])"; done
