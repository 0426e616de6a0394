#!/bin/bash
# scripts/build_variant.sh <tag> [-D...]  -> raytracer_challenge_amd/csrc/variants/librtc_amd_<tag>.so (objects in variants/obj_<tag>/)
cd /root/repo/raytracer_challenge_amd/csrc || exit 1
tag=$1; shift
mkdir -p variants/obj_$tag
make -s O=variants/obj_$tag OUT=variants/librtc_amd_$tag.so DEFS="$*" 2>&1 | grep -E "error" 
ls -la variants/librtc_amd_$tag.so | awk '{print $5, $9}'
