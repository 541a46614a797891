#!/bin/bash
cd "$(dirname "$0")/../.."
P="python tools/dev/zoo_shape.py"
$P 16 16 3 2 16384 2>&1 | grep -v amdgpu
CT_FLAGS=2 $P 16 16 3 2 16384 2>&1 | grep -v amdgpu
CLOUDCT_SPLAT_BWD_NSEG=2 $P 16 16 3 2 16384 2>&1 | grep -v amdgpu
CLOUDCT_SPLAT_BWD_NSEG=4 $P 16 16 3 2 16384 2>&1 | grep -v amdgpu
CT_FLAGS=2 CLOUDCT_SPLAT_BWD_NSEG=2 $P 16 16 3 2 16384 2>&1 | grep -v amdgpu
$P 16 64 2 2 16384 2>&1 | grep -v amdgpu
CLOUDCT_SPLAT_BWD_NSEG=2 $P 16 64 2 2 16384 2>&1 | grep -v amdgpu
CLOUDCT_SPLAT_BWD_NSEG=4 $P 16 64 2 2 16384 2>&1 | grep -v amdgpu
$P 32 8 3 2 16384 2>&1 | grep -v amdgpu
CLOUDCT_SPLAT_BWD_NSEG=2 $P 32 8 3 2 16384 2>&1 | grep -v amdgpu
CLOUDCT_SPLAT_BWD_NSEG=4 $P 32 8 3 2 16384 2>&1 | grep -v amdgpu
