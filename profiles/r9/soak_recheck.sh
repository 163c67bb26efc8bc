cd /root/repo; PKG=diff-gaussian-rasterization_amd
for lib in prev cur prev cur; do
cp $PKG/lib_$lib/libdgr_hip.so $PKG/lib/libdgr_hip.so
echo "== $lib"
DGR_SOAK_ONLY=246 timeout 600 python tests/tools/soak_parity.py 250 0 81 2>&1 | grep -v amdgpu.ids | grep "FAIL\|draws in" | cut -c1-330
DGR_SOAK_HEAVY=1 DGR_DETERMINISTIC_GRADS=1 DGR_SOAK_ONLY=132 timeout 600 python tests/tools/soak_parity.py 150 0 85 2>&1 | grep -v amdgpu.ids | grep "FAIL\|draws in" | cut -c1-330
done
cp $PKG/lib_cur/libdgr_hip.so $PKG/lib/libdgr_hip.so
