cd $GRAFT_REPO_ROOT
python tools/seed_iter_probe.py 2>&1 | grep -v "amdgpu.ids\|bench\]"
