cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp GPU_MAX_HW_QUEUES=24
for w in 2 4; do SELFTEST_MODE=raw timeout -k 10 300 python3 tests/fake_rccl/selftest.py $w 150 2>&1 | grep -v "^W2026\|amdgpu.ids" | tail -14; done
