#!/bin/bash
# Why is the data-parallel process slower than the plain one at one rank?  (3) hardware-queue aliasing of the library's side stream?
cd "$GRAFT_REPO_ROOT"; out=gpurun_out/ddp_slow3.txt; : > $out
b() { "$@" 2>/dev/null | tail -1 | grep -o '"ms_per_step": [0-9.]*'; }
A="bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-extras"
for i in 1 2; do
echo "plain                                          $(b python $A)" | tee -a $out
echo "process group created, unused                  $(b python $A --init-pg-only)" | tee -a $out
echo "  + GPU_MAX_HW_QUEUES=8                        $(GPU_MAX_HW_QUEUES=8 b python $A --init-pg-only)" | tee -a $out
echo "  + GPU_MAX_HW_QUEUES=2                        $(GPU_MAX_HW_QUEUES=2 b python $A --init-pg-only)" | tee -a $out
echo "  + TORCH_NCCL_ENABLE_MONITORING=0             $(TORCH_NCCL_ENABLE_MONITORING=0 b python $A --init-pg-only)" | tee -a $out
echo "  + single stream                              $(HSIMAE_TWO_STREAMS=0 b python $A --init-pg-only)" | tee -a $out
echo "plain, GPU_MAX_HW_QUEUES=8                     $(GPU_MAX_HW_QUEUES=8 b python $A)" | tee -a $out
echo "--force-ddp, GPU_MAX_HW_QUEUES=8               $(GPU_MAX_HW_QUEUES=8 RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 b python $A --force-ddp)" | tee -a $out
done
