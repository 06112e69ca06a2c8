echo "== adam: failed swaps dropped"; CYMF_RELMF_TILE_DBG=8 timeout -k 10 120 python3 tools/relmf_check.py quality 2>&1 | grep "adam" | cut -c1-160
echo "== 4 waves"; CYMF_RELMF_TILE_WAVES=4 timeout -k 10 120 python3 tools/relmf_check.py quality 2>&1 | grep "adam" | cut -c1-160
echo "== 4 waves, B 187"; CYMF_RELMF_TILE_B=187 CYMF_RELMF_TILE_WAVES=4 timeout -k 10 120 python3 tools/relmf_check.py quality 2>&1 | grep "adam" | cut -c1-160
echo "== old step path"; CYMF_RELMF_NO_TILES=1 timeout -k 10 120 python3 tools/relmf_check.py quality 2>&1 | grep "adam" | cut -c1-160
