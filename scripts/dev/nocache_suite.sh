#!/bin/bash
# Dev, on the GPU box: the GPU suite and a few bench shapes with PyTorch's caching allocator switched off.  Every tensor is then its own
# hipMalloc, freed memory is really unmapped and a read past an allocation faults instead of landing in a neighbour: out-of-bounds reads
# and use-after-free across streams become deterministic (found nothing after the gemm_tn fix of round 5; the hipGraph capture test is
# excluded: capture cannot allocate without the caching allocator).
set -u
cd "${GRAFT_REPO_ROOT:-$(pwd)}"; ulimit -c 0
export PYTORCH_NO_CUDA_MEMORY_CACHING=1 PYTORCH_NO_HIP_MEMORY_CACHING=1
timeout 2400 python -m pytest tests -m gpu -q -k "not graph" 2>&1 | grep -E "passed|failed|^FAILED" | tail -5
for a in "" "--variant global_local" "--variant region_mem" "--prune-top" "--frames 4" "--batch 24 --frames 3"; do
  timeout 300 python bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-other-configs --no-traffic --no-forced-w1 $a > /tmp/o.txt 2> /tmp/e.txt
  echo "bench [$a] rc=$? lines=$(grep -c '^{' /tmp/o.txt) $(grep -i -m1 'illegal\|fault' /tmp/e.txt | cut -c1-100)"
done
