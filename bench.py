#!/usr/bin/env python3
"""bench.py -- headline benchmark: Pippenger MSM scalar-point pairs/s at n = 2^20 over
secp256k1 on MI355X, plus the second half of BASELINE.json's metric (range-proof verifies/s,
config C5) and the inner-product-argument prover of config C3 as `extra`.

  python bench.py [--gpus N] [--steps K] [--warmup W] [--logn 20] [--scaling weak|strong]

`--gpus N` with N > 1 and no WORLD_SIZE in the environment: this process touches no GPU; it
starts `python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py <same args>`
as a CHILD process (one rank per GPU over RCCL), relays rank 0's JSON line and exits with the
child's code.  Under torch.distributed.run (the driver's own launch form) it is a rank.

One "step" = one MSM of n pairs per GPU with scalars and points already resident in HBM; the
64-byte result comes back to the host every step.  Steps are pipelined two deep through
bpmi_msm_dev_enqueue / bpmi_msm_finish: the host tail of step k (and, with N > 1, the
all_gather of the 64-byte partials + bpmi_ec_sum fold) overlaps the kernels of step k + 1;
all K results are complete inside the timed region.
  weak   (default): every rank owns its own n pairs; the global MSM has N*n pairs.
  strong: ONE MSM of n pairs split N ways (n/N pairs per rank), north_star's "large MSMs
          shard across the GPUs of one node".

Prints ONE JSON line on rank 0 (driver contract), with
  roofline     -- the dominant kernel (k_accum_l0) against the HBM roofline that north_star
                  prescribes: algorithmic bytes = 96 B/pair, duration from HIP events on the
                  launch stream inside the timed region;
  alu_roofline -- the honest utilisation figure for this integer path: multiply-add lane
                  operations per second against the raw v_mad_u64_u32 rate of the chip;
  cpu_baseline -- the plain-C oracle MSM ("port") on this host's cores, bounded sample;
  result_ok    -- the timed MSM's 64 bytes against the known answer (sum e_i k_i) * G;
  extra        -- C5 batch verification (verifies/s) and C3 IPA prover (seconds), each with its
                  own roofline object.
"""
import os
import sys

REPO = os.path.dirname(os.path.abspath(__file__))
if REPO not in sys.path:
    sys.path.insert(0, REPO)
# (benchlib.common sets GPU_MAX_HW_QUEUES before anything imports torch: the two MSM lanes need hardware queues of their own)
from benchlib.headline import main  # noqa: E402

# the pieces, re-exported for callers that imported them from here (tests/, tools/)
from benchlib.common import *  # noqa: E402,F401,F403
from benchlib.cpu_baseline import c5_cpu_baseline, cpu_baseline, cpu_reference_algorithm  # noqa: E402,F401
from benchlib.extras import extra_c2, extra_c3, extra_c4, extra_c5  # noqa: E402,F401
from benchlib.launch import PeerFailure, Ready, c5_inflight, free_port, launch_ranks, run_extras  # noqa: E402,F401

if __name__ == "__main__":
    main()
