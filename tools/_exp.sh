for d in 0 2 3; do echo "DBG=$d"; MEVI_H1_DBG=$d timeout 200 python tools/probe_dense.py 4000000 2>&1 | tail -1; done
