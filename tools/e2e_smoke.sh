# End-to-end smoke of the harness on one GPU: real token shards (the reference's directory layout), dropout on,
# checkpoints, resume.  Run from the repo root on a GPU box: bash tools/e2e_smoke.sh
set -e -o pipefail
cd /root/repo
python - <<'PY'
import sys, os
sys.path.insert(0, "oracle")
from gen_golden_loader import make_shards
base = "gpurun_out/e2e_data"
for d in ("genbank/train", "uniref100/train", "genbank/val", "uniref100/val"):
    os.makedirs(os.path.join(base, d), exist_ok=True)
    make_shards(os.path.join(base, d), 6, seed=hash(d) % 1000)
print("shards ok")
PY
rm -rf /tmp/e2e_run && mkdir -p /tmp/e2e_run && cd /tmp/e2e_run   # checkpoints are ~70 MB each: keep them out of gpurun_out
python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29517 /root/repo/training/train_encoder.py --batch_size 32 --mini_batch_size 4 --ctx_len 256 --n_layer 2 --n_embd 256 --n_head 2 --dropout 0.1 --base_dir /root/repo/gpurun_out/e2e_data --max_steps 8 --save_freq 4096 --test_freq 20000 --save_name e2e 2>&1 | tail -12
ls
python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29517 /root/repo/training/train_encoder.py --batch_size 32 --mini_batch_size 4 --ctx_len 256 --n_layer 2 --n_embd 256 --n_head 2 --dropout 0.1 --base_dir /root/repo/gpurun_out/e2e_data --max_steps 4 --save_freq 4096 --save_name e2e --resume_from 65536 2>&1 | tail -8
