#!/bin/bash
# lab builds of the library for tools/leak_probe.py (no GPU needed to build)
set -e
cd "$(dirname "$0")/.."
mkdir -p tmp_exp/leak
F="-Wno-pass-failed -Wno-int-to-void-pointer-cast -O3 -std=c++17 --offload-arch=gfx950 -fPIC -shared -fno-slp-vectorize"
/opt/rocm/bin/hipcc $F -o tmp_exp/leak/libsdrxgpu_product.so sdrx_amd/csrc/sdrxgpu.hip &
/opt/rocm/bin/hipcc $F -DSDRX_LAB_UPLOAD=1 -o tmp_exp/leak/libsdrxgpu_null_upload.so sdrx_amd/csrc/sdrxgpu.hip &
/opt/rocm/bin/hipcc $F -DSDRX_LAB_UPLOAD=1 -DSDRX_LAB_CREATE_SYNC=1 -o tmp_exp/leak/libsdrxgpu_null_devsync.so sdrx_amd/csrc/sdrxgpu.hip &
/opt/rocm/bin/hipcc $F -DSDRX_LAB_UPLOAD=1 -DSDRX_LAB_CREATE_SYNC=2 -o tmp_exp/leak/libsdrxgpu_null_nullsync.so sdrx_amd/csrc/sdrxgpu.hip &
wait
ls -la tmp_exp/leak
