#!/bin/bash
# usage: tools/dev/isa.sh <mangled-name-regex> : dumps the gfx950 ISA of one raster kernel to /tmp/k.s
cd /tmp && /opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -std=c++17 -ffp-contract=off -I /root/repo/include -S --cuda-device-only /root/repo/cloud_transformers_amd/csrc/ct_raster.hip -o raster.s 2>/dev/null
name=$(grep -o "^_Z[A-Za-z0-9_]*:" raster.s | tr -d ':' | grep -E "$1" | head -1)
echo "kernel: $name"
awk -v n="$name:" '$1==n {p=1} p {print} p && /s_endpgm/ {exit}' raster.s > k.s
wc -l k.s
