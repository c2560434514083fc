for e in 2 3 4 5 6 7 8 9 10 11 12 13 14 15 16 17 18 19 20; do python tools/real_pair_loop.py $e 15 2>&1 | grep "loop wall" | sed "s/^/epoch $e: /"; done
