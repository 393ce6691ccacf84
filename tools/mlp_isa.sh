#!/bin/bash
# compile codec_gemm.hip to ISA and summarise the fused MLP kernel(s): registers, scratch, compiler-inserted vmcnt waits
cd /root/repo/chattts_amd/csrc
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-gpu-rdc -Wno-unused-result -S --cuda-device-only codec_gemm.hip -o /tmp/cg.s -Rpass-analysis=kernel-resource-usage 2>&1 | grep -A9 "Name: _Z[0-9]*mlp_fused" | grep -E "Name|SGPRs:|VGPRs:|Scratch|Occupancy|LDS"
for k in $(grep -o "^_Z[0-9]*mlp_fused[A-Za-z0-9_]*:" /tmp/cg.s | tr -d ':'); do
  awk "/^$k:/,/s_endpgm/" /tmp/cg.s > /tmp/$k.s
  echo "$k: $(grep -c v_mfma /tmp/$k.s) mfma, $(grep -c scratch_ /tmp/$k.s) scratch ops; vmcnt waits:"; grep "s_waitcnt vmcnt" /tmp/$k.s | awk '{print $2}' | sort | uniq -c | tr '\n' ' '; echo
done
