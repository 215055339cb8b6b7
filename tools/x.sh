for d in 0 1 2 4 8 3 7 15; do echo "DBG=$d"; MSSVT_DBG=$d python tools/prof_attn.py 2>&1 | grep -o "k_ffn[^}]*avg_launch_us': [0-9.]*" | grep -o "avg_launch_us.*"; done
