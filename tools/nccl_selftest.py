"""Exercises the RCCL path of bulletproofs_amd.distributed on however many GPUs torchrun
gives it (also world_size 1): all_gather of 64-byte partials + GPU fold must equal the MSM
over the concatenated shards."""
import os, sys, random
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import torch, torch.distributed as dist
import bulletproofs_amd
from bulletproofs_amd.engine import Engine
from bulletproofs_amd.distributed import ShardedMSM, all_gather_bytes
from bulletproofs_amd.ec import secp256k1

local = int(os.environ.get("LOCAL_RANK", "0"))
torch.cuda.set_device(local)
os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
dist.init_process_group("nccl", device_id=torch.device("cuda", local))
rank, world = dist.get_rank(), dist.get_world_size()
eng = Engine(device=local)
Q = secp256k1.q
rnd = random.Random(5)            # same on every rank
n = 4096
G = secp256k1.G.to_le64()
ks = [rnd.randrange(1, Q) for _ in range(n * world)]
es = [rnd.randrange(Q) for _ in range(n * world)]
pack = lambda v: b"".join(x.to_bytes(32, "little") for x in v)
allpts = eng.ec_mul_batch_bytes(G * (n * world), pack(ks), n * world)
sm = ShardedMSM(engine=eng)
lo, hi = rank * n, (rank + 1) * n
part = eng.msm_bytes(allpts[64 * lo:64 * hi], pack(es[lo:hi]), n)
parts = all_gather_bytes(part)
assert parts[rank] == part and len(parts) == world
got = eng.ec_sum_bytes(b"".join(parts), world)
want = eng.msm_bytes(allpts, pack(es), n * world)
assert got == want, "sharded MSM != full MSM"
assert sm.combine(part) == want, "ShardedMSM.combine (device-side fold) != full MSM"
d = eng.upload(b"".join(parts))
assert eng.ec_sum_dev(d, world) == want
kg = (sum(e * k for e, k in zip(es, ks)) % Q)
assert want == eng.ec_mul_batch_bytes(G, kg.to_bytes(32, "little"), 1)
dist.barrier()
if rank == 0:
    print("NCCL_SELFTEST_OK world=%d" % world)
dist.destroy_process_group()
