cd $GRAFT_REPO_ROOT
for l in 3 4; do python tools/profile_block.py --level $l --iters 30 2>&1 | grep -v amdgpu; python tools/profile_block.py --level $l --iters 30 --decoder 1 2>&1 | grep -v amdgpu; done
for l in 1 2; do python tools/profile_block.py --level $l --iters 30 2>&1 | grep -v amdgpu; done
