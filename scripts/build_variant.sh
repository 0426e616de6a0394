#!/bin/bash
# scripts/build_variant.sh <tag> [-D...]  -> raytracer_challenge_amd/csrc/variants/librtc_amd_<tag>.so  (+ resource usage line)
cd /root/repo/raytracer_challenge_amd/csrc || exit 1
mkdir -p variants
tag=$1; shift
F="--offload-arch=gfx950 -std=c++17 -O3 -fPIC -ffp-contract=off -fno-fast-math -Wno-unused-function -Wno-unused-variable"
/opt/rocm/bin/hipcc $F "$@" -shared -o variants/librtc_amd_$tag.so rtc_kernels.hip rtc_scene.cpp rtw_capi.cpp 2>&1 | grep -E "error"
/opt/rocm/bin/hipcc $F "$@" -Rpass-analysis=kernel-resource-usage -c rtc_kernels.hip -o /dev/null 2>&1 | grep -A9 "ILb0" | grep -E " VGPRs:|Scratch|Occupancy|LDS" | sed 's/.*remark: *//; s/ \[-R.*//' | tr '\n' ';'
echo " <- $tag"
