#!/bin/bash
# Diagnostic library with per-workgroup phase stamps in the gate / dgrad kernel (tools/clockprobe.py).
set -e
cd "$(dirname "$0")/.."
python nasa-niswan_amd/build.py > /dev/null
/opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -fPIC -std=c++17 -Iinclude -Inasa-niswan_amd/csrc -DNINT_STAMP \
    -c nasa-niswan_amd/csrc/conv_igemm.hip -o nasa-niswan_amd/build/conv_igemm_stamp.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o nasa-niswan_amd/build/libnint_stamp.so \
    nasa-niswan_amd/build/conv_igemm_stamp.o nasa-niswan_amd/build/wgrad.o nasa-niswan_amd/build/pointwise.o nasa-niswan_amd/build/seq.o
echo nasa-niswan_amd/build/libnint_stamp.so
