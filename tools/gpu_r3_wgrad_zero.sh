cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
for z in 0 1 0 1; do echo "== zero=$z"; WG_ZERO=$z BIN_AMD_LIB=tools/_abl/libbinhip_tuning.so WG_DBGS=0,0x10000 WG_LAYERS="3,96,32;3,160,32;3,192,32" timeout 200 python tools/bench_wgrad.py 2>&1 | grep -v amdgpu.ids; done | tee gpurun_out/r3_wgrad_zero.log
