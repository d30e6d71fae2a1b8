for s in 0 4 8 12 16 24; do echo "stagger=$s"; HVC_GEMM_STAGGER=$s python scripts/gemm_vs_blas.py 2>&1 | grep "fwd\|dx" | awk '{print "   ", $0}'; done
