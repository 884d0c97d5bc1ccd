cd $GRAFT_REPO_ROOT
python tools/diag_ckpt.py 2>&1 | grep -v amdgpu.ids
echo "=== SWF_WIN24=0"
SWF_WIN24=0 python tools/diag_ckpt.py 2>&1 | grep -v amdgpu.ids
