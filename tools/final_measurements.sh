PMC_SUMMARY=profiles/r05_k_pmc_dense_spmm.txt tools/gpu_session.sh pmc benchworld2 benchbig
PMC_ARGS="--shape 1m-500k --emb 128" PMC_NAME=_1m-500k_d128 PMC_TAIL=30 SPMM_PMC_LAUNCHES=30 PMC_TIMEOUT=600 PMC_SUMMARY=profiles/r05_l_pmc_dense_spmm_1m500k.txt PMC_WHAT="spmm_rows_kernel<32,false>, dense value-free flavour, 1m-500k-shape graph, d=128" tools/gpu_session.sh pmc
