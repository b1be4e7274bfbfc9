# Bytes on the wire of the gradient exchange per rank and step, by camera rig and exchange: N ranks over gloo on ONE GPU (the collectives are
# real, the links are not -- no timing is taken from this).  gpurun -- 'bash tools/exchange_bytes.sh' -> gpurun_out/exchange_bytes.json
export SURFEL_DIST_BACKEND=gloo SURFEL_ALLOW_SINGLE_CHANNEL=1
mkdir -p gpurun_out; : > gpurun_out/exchange_bytes.jsonl
port=29611
for n in 2 8; do for rig in benchmark inside; do for ex in allreduce compacted; do
  port=$((port+1))
  timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node $n --master-addr 127.0.0.1 --master-port $port bench.py --gpus $n --steps 2 --warmup 1 \
      --no-cpu-baseline --no-train-step --exchange $ex --rig $rig 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); c=d['config']
print(json.dumps({'n_ranks': d['n_gpus'], 'rig': c['camera_rig'], 'exchange': '$ex', 'gaussians': c['gaussians'], 'visible_per_P_rank0': c['visible_per_P'],
  'rows_union_over_P': c.get('exchange_rows_union_over_P'), 'bytes_payload_per_step_rank0': c['exchange_bytes_sent_per_step_rank0'],
  'bytes_dense_allreduce': c['exchange_bytes_dense_allreduce_equivalent'], 'selfcheck': c['exchange_selfcheck']}))" >> gpurun_out/exchange_bytes.jsonl
done; done; done
python -c "
import json; rows=[json.loads(l) for l in open('gpurun_out/exchange_bytes.jsonl') if l.strip()]
json.dump({'what': 'payload bytes per rank and step of the gradient exchange (3 M Gaussians, 232 B of gradients each), N ranks over gloo on one GPU', 'runs': rows}, open('gpurun_out/exchange_bytes.json','w'), indent=1)
for r in rows: print(r)"
