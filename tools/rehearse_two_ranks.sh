set -e
cd $GRAFT_REPO_ROOT
python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 2 --steps 5 --warmup 1 --restarts 6 --backend gloo --share-gpu --no-cpu-baseline 2> gpurun_out/r02j_rehearse2.err | tail -1 > gpurun_out/r02j_rehearse2.json
python -c "
import json; d=json.load(open('gpurun_out/r02j_rehearse2.json')); print(d['n_gpus'], d['value'], d['scaling'], d['config']['best_restart'], d['loop_only'])"
