#!/bin/bash
# The HOST side of libcrt_amd.so (builder, scene flattening, image check, C ABI, the renderer's host code) under
# AddressSanitizer + UndefinedBehaviorSanitizer, on the CPU (no GPU needed; GPU sanitizers are not available on this
# pool): device code is compiled as usual (-fno-gpu-sanitize), the host objects are instrumented, and the CPU tests
# that drive the host code run against the instrumented library.   bash profiles/host_sanitizers.sh
set -e
R=$(cd "$(dirname "$0")/.." && pwd); T=${TMPDIR:-/tmp}/crt_asan; mkdir -p $T; cd $R/crust-render_amd/csrc
F="--offload-arch=gfx950 -g -fPIC -std=c++17 -ffp-contract=off -fno-fast-math -I../../include -fsanitize=address,undefined -fno-gpu-sanitize -fno-omit-frame-pointer"
for f in bvh_build scene capi; do /opt/rocm/bin/hipcc $F -O1 -x hip -c $f.cpp -o $T/$f.o 2>/dev/null & done
for f in traverse pathtrace shade_seam gather; do /opt/rocm/bin/hipcc $F -O2 -fno-slp-vectorize -c kernels/$f.hip -o $T/$f.o 2>/dev/null & done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -fsanitize=address,undefined -fno-gpu-sanitize -o $T/libcrt_asan.so $T/*.o -lpthread
cd $R
RT=$(find /opt/rocm/lib/llvm -name "libclang_rt.asan-x86_64.so" | head -1)
CRT_AMD_LIB=$T/libcrt_asan.so LD_PRELOAD=$RT ASAN_OPTIONS=detect_leaks=0:abort_on_error=0 UBSAN_OPTIONS=print_stacktrace=1 \
  python -m pytest tests/test_build_parity.py tests/test_abi.py tests/test_stress_scene.py -q -p no:cacheprovider 2>&1 | tee $T/run.log | tail -3
echo "ASan / UBSan reports: $(grep -c 'runtime error\|ERROR: AddressSanitizer' $T/run.log)"
# ... and under ThreadSanitizer: the builder forks subtrees onto helper threads (bvh_build.cpp); the parity and stress
# scene tests commit trees large enough to fork (43 200 triangles, the 382 804-triangle stress scene, MedCity)
T2=${TMPDIR:-/tmp}/crt_tsan; mkdir -p $T2; cd $R/crust-render_amd/csrc
F2="--offload-arch=gfx950 -g -fPIC -std=c++17 -ffp-contract=off -fno-fast-math -I../../include -fsanitize=thread -fno-gpu-sanitize -fno-omit-frame-pointer"
for f in bvh_build scene capi; do /opt/rocm/bin/hipcc $F2 -O1 -x hip -c $f.cpp -o $T2/$f.o 2>/dev/null & done
for f in traverse pathtrace shade_seam gather; do /opt/rocm/bin/hipcc $F2 -O2 -fno-slp-vectorize -c kernels/$f.hip -o $T2/$f.o 2>/dev/null & done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -fsanitize=thread -fno-gpu-sanitize -o $T2/libcrt_tsan.so $T2/*.o -lpthread
cd $R
RT2=$(find /opt/rocm/lib/llvm -name "libclang_rt.tsan-x86_64.so" | head -1)
CRT_AMD_LIB=$T2/libcrt_tsan.so LD_PRELOAD=$RT2 TSAN_OPTIONS="halt_on_error=0" \
  python -m pytest tests/test_build_parity.py tests/test_stress_scene.py -q -p no:cacheprovider 2>&1 | tee $T2/run.log | tail -3
echo "TSan reports: $(grep -c 'WARNING: ThreadSanitizer' $T2/run.log)"
