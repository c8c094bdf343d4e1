#!/bin/bash
# The scaling curve on a multi-GPU node (the driver's 8-GPU box):   bash tools/run_scale.sh [outdir] [ngpu-list] [total-nsub]
#
#   weak    bench.py --gpus N                      every rank fits its own 1024 subints per step (BASELINE's metric)
#   strong  bench.py --gpus N --total-nsub 100000  configs[4] as written: contiguous shards, one gather
#
# for N in 1 2 4 8 (as many as the node has), one rank per GPU over RCCL, launched the way the driver launches
# bench.py.  Every run's return code, row count and checksums are checked against the N = 1 records
# (tools/check_scale.py): a subint's answer is a function of that subint alone, so the strong job's gathered
# records must be THE SAME BITS for every N, however the shards and sub-batches fall.  Writes one table,
# <outdir>/scale_table.txt.  Nothing here computes an efficiency figure for the judge: the driver does that itself.
out=${1:-gpurun_out/scale}
ngpus=${2:-"1 2 4 8"}
total=${3:-100000}
steps=${STEPS:-10}
warm=${WARMUP:-3}
mkdir -p "$out"
have=$(python -c 'import torch; print(torch.cuda.device_count())')
export HSA_ENABLE_IPC_MODE_LEGACY=0
port=29611
status=0
for n in $ngpus; do
  if [ "$n" -gt "$have" ]; then echo "skip N=$n: the node has $have GPU(s)" | tee -a "$out/scale_table.txt"; continue; fi
  for mode in weak strong; do
    extra="--steps $steps --warmup $warm --no-other-workloads --no-cpu-baseline"
    [ "$mode" = strong ] && extra="--total-nsub $total --no-cpu-baseline --dump-records $out/records_${mode}_n$n.npy"
    port=$((port + 1))
    # the plain command for every N: bench.py starts its own ranks (torch.distributed.run as a child process) when
    # --gpus N > 1 and no launcher has set RANK; LAUNCHER=torchrun uses the driver's documented form instead
    if [ "$n" -eq 1 ] || [ "${LAUNCHER:-plain}" = plain ]; then
      MASTER_PORT=$port python bench.py --gpus "$n" $extra > "$out/${mode}_n$n.json" 2> "$out/${mode}_n$n.err"
    else
      python -m torch.distributed.run --nnodes=1 --nproc-per-node "$n" --master-addr 127.0.0.1 --master-port $port \
        bench.py --gpus "$n" $extra > "$out/${mode}_n$n.json" 2> "$out/${mode}_n$n.err"
    fi
    rc=$?
    echo "$mode N=$n rc=$rc" >> "$out/rc.txt"
    [ $rc -ne 0 ] && status=1
  done
done
python tools/check_scale.py "$out" $ngpus | tee "$out/scale_table.txt" || status=1
exit $status
