#!/bin/bash
# Host side of libexpv_mi.so under AddressSanitizer (or, with SAN=undefined, UBSan): a separate build in /tmp/asan, never loaded by
# the product.  Run the CPU ABI tests against it:
#   bash tools/build_sanitized.sh && ASAN_OPTIONS=detect_leaks=0:halt_on_error=0:log_path=/tmp/asan/log \
#     LD_PRELOAD=/opt/rocm/lib/llvm/lib/clang/22/lib/linux/libclang_rt.asan-x86_64.so EXPV_MI_LIB=/tmp/asan/libexpv_mi_asan.so \
#     python -m pytest tests/test_abi_cpu.py tests/test_oracle_c.py -q -m "not gpu"
# (-O1: the two bit-for-bit tests of the symmetric tridiagonal eigen-solver compare two template instantiations and need the product's -O3)
mkdir -p /tmp/asan && cd /tmp/asan
SAN=${SAN:-address}
set -e
CS=/root/repo/exponentialutilities.jl_amd/csrc
FL="--offload-arch=gfx950 -O1 -g -std=c++17 -fPIC -Wno-unused-function -Xarch_host -mavx2 -Xarch_host -mfma -Xarch_host -fcx-limited-range -Xarch_host -fsanitize=$SAN -Xarch_host -fno-omit-frame-pointer"
for s in kernels fused pipe engine_core engine_drivers engine_batch capi; do
  ( /opt/rocm/bin/hipcc $FL -c $CS/$s.hip -o $s.o 2> $s.err || echo FAIL $s ) &
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -shared-libsan -fsanitize=$SAN -o libexpv_mi_${SAN/address/asan}.so kernels.o fused.o pipe.o engine_core.o engine_drivers.o engine_batch.o capi.o
ls -la libexpv_mi_${SAN/address/asan}.so
