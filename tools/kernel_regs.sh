#!/bin/bash
# Register / LDS / scratch use of the kernels of one translation unit (cross-compiles, no GPU):
#   tools/kernel_regs.sh wl_strip_hip.hip [filter-regex] [extra hipcc flags...]
src=$1; pat=${2:-.}; shift; shift
out=${TMPDIR:-/tmp}/wl_regs_$$
mkdir -p $out
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fno-strict-aliasing -fPIC -fno-slp-vectorize -Wno-inline-asm \
    --cuda-device-only -S "$@" /root/repo/pytorch_wavelets_amd/csrc/$src -o $out/k.s || exit 1
awk '/^[ \t]*\.amdhsa_kernel[ \t]/{k=$2} /\.amdhsa_next_free_vgpr/{v=$2} /\.amdhsa_next_free_sgpr/{s=$2} /\.amdhsa_group_segment_fixed_size/{l=$2} /\.amdhsa_private_segment_fixed_size/{p=$2} /\.end_amdhsa_kernel/{print k, "vgpr", v, "sgpr", s, "lds", l, "scratch", p}' $out/k.s \
  | sed -e 's/^_Z9wl_kernelI//' | grep -E "$pat"
rm -rf $out
