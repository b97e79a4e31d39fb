for k in 0 1 2 4 7; do echo "== knock $k"; WOQ_PERSIST_KNOCK=$k WOQ_PERSIST_HINTS=0 timeout 100 python tools/persist_stamps.py 8 2>&1 | grep -E "^(qkv|o |gate|down|layers)" ; done
