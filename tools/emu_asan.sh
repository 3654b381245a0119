#!/bin/bash
# CPU container: the kernel sources on the fiber emulator, built with host AddressSanitizer and with every workgroup's
# LDS poisoned (tests/emu/hip_emu.h: poison_lds), through the emulator parity tests.  This is the sanitizer pass of the
# device code that exists in this image: a device-side sanitizer build compiles, but its
# runtime cannot allocate device memory on the GPU boxes (no /opt/rocm/lib/asan; profiles/r05_a_fault_rootcause.txt).
#   bash tools/emu_asan.sh [pytest -k expression]      -> build/emu_asan.log
set -u
cd "$(dirname "$0")/.."
mkdir -p build
g++ -std=c++17 -O1 -g -fPIC -fdata-sections -fsanitize=address -fno-omit-frame-pointer -shared -DJPP_EMU -Itests/emu -Ijumanpp_amd/csrc \
    -x c++ jumanpp_amd/csrc/jppgpu_api.cc -o build/libjppgpu_emu_asan.so -Wl,-T,tests/emu/lds.ld || exit 1
K="${1:-not cli and not host and not train}"
LD_PRELOAD="$(gcc -print-file-name=libasan.so)" ASAN_OPTIONS=detect_leaks=0 JPPEMU_TEST_LIB="$PWD/build/libjppgpu_emu_asan.so" \
  timeout 3000 python -m pytest tests/test_cpu_parity.py tests/test_ref_fixtures.py tests/test_scorers.py -x -q -m "not gpu" -k "$K" > build/emu_asan.log 2>&1
echo "rc $?"; grep -v "^  File" build/emu_asan.log | tail -15
