#!/bin/bash
# Evidence for the batch-preparation kernel (config C5, one GPU; run from the repo root ON THE GPU BOX):
#   bash tools/c5_prepare_sweep.sh > gpurun_out/c5_prepare.txt
# host vs device preparation, pageable vs page-locked receive buffer, proofs per wave, and each role alone.
P="python3 tools/profile_c5.py"
F='rep 2\|rp_prepare\|rp_elements\|ec_decompress'
echo "== preparation on the host (bpmi_rp_batch_prepare, 32 threads), bytes receive buffer"; C5_PREPARE=host $P 2>&1 | grep "rep 2"
echo "== preparation on the device, bytes (pageable) receive buffer, commitments as Point objects"; C5_PREPARE=device $P 2>&1 | grep "$F"
echo "== preparation on the device, page-locked receive buffer, commitments packed"; C5_PINNED=1 C5_PREPARE=device $P 2>&1 | grep "$F"
for l in 8 16 32 64; do echo "== proofs per wave (rp_lanes) = $l"; C5_LANES=$l C5_PINNED=1 C5_PREPARE=device $P 2>&1 | grep "$F"; done
echo "== stages one after the other (rp_overlap = 0: the point decoding BEHIND the preparation kernels)"; C5_SERIAL=1 C5_PINNED=1 C5_PREPARE=device $P 2>&1 | grep "$F"
for r in 0 1 2 3; do echo "== only role $r of k_rp_roles (0: Protocol-2 hash chain, 1: the other transcript checks, 2: inversion + tables, 3: inverse-free scalars); not a verification"; C5_SERIAL=1 C5_ONLY_ROLE=$r C5_PINNED=1 C5_PREPARE=device $P 2>&1 | grep "rp_prepare\|rp_elements"; done
