cd $GRAFT_REPO_ROOT
for v in 0 1; do CHEBGCN_ORD_SMALL=$v python tools/probes/n1000_grad_noise.py 1000 3 2>&1 | grep seed; done
