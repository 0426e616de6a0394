#!/bin/bash
# GPU box: counter list of this rocprofv3 + VALU / thread-utilisation PMC passes of the current kernels (config 2 wavefront, config 3 one kernel).
export TMPDIR=/tmp
mkdir -p gpurun_out/r2a
rocprofv3 -L > gpurun_out/r2a/counters_list.txt 2>&1
for WL in config2 config3; do
  B="python3 bench.py --steps 4 --warmup 1 --no-cpu-baseline --extra-workloads= --workload $WL"
  i=0
  for C in "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES SQ_INSTS_SALU SQ_INSTS_VMEM" \
           "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_LDS SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS"; do
    i=$((i+1))
    timeout -k 10 200 rocprofv3 --kernel-trace --pmc $C --output-format csv -d gpurun_out/r2a/pmc_${WL}_$i -- $B > gpurun_out/r2a/pmc_${WL}_$i.log 2>&1 || echo "pass $WL $i failed" >> gpurun_out/r2a/errors.txt
  done
done
python3 scripts/pmc_by_kernel.py gpurun_out/r2a/pmc_* > gpurun_out/r2a/by_kernel.txt 2>&1
tail -60 gpurun_out/r2a/by_kernel.txt
