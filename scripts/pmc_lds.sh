#!/bin/bash
# Runs ON THE GPU BOX: LDS / VMEM pressure counters of gemm_nt alone (scripts/pmc_gemm_loop.py), two separate --pmc passes.
set -u
cd "${GRAFT_REPO_ROOT:-$(pwd)}"; export TMPDIR=/tmp
OUT=$PWD/gpurun_out/pmc_lds; rm -rf "$OUT"; mkdir -p "$OUT"
timeout 300 rocprofv3 --kernel-trace --pmc SQ_BUSY_CU_CYCLES SQ_ACTIVE_INST_LDS SQ_LDS_CMD_FIFO_FULL SQ_LDS_DATA_FIFO_FULL SQ_ACTIVE_INST_VMEM --output-format csv -d "$OUT/p1" -o run -- python scripts/pmc_gemm_loop.py > "$OUT/p1.log" 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc SQ_BUSY_CU_CYCLES SQ_INST_CYCLES_VMEM_RD SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_INSTS_LDS --output-format csv -d "$OUT/p2" -o run -- python scripts/pmc_gemm_loop.py > "$OUT/p2.log" 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc SQ_BUSY_CU_CYCLES SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_VALU_MFMA_BUSY_CYCLES --output-format csv -d "$OUT/p3" -o run -- python scripts/pmc_gemm_loop.py > "$OUT/p3.log" 2>&1
python - <<'PY'
import glob, pandas as pd
for p in ("p1", "p2", "p3"):
    fs = glob.glob(f"gpurun_out/pmc_lds/{p}/**/*counter_collection.csv", recursive=True)
    if not fs:
        print(p, "no counters", open(f"gpurun_out/pmc_lds/{p}.log").read()[-400:]); continue
    d = pd.read_csv(fs[0])
    d = d[d.Kernel_Name.str.contains("gemm_nt_kernel<0, 2, 4, 8")]
    t = d.groupby(["Dispatch_Id", "Counter_Name"]).Counter_Value.sum().unstack()
    print(p); print(t.to_string())
PY
