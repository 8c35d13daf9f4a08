lib=racing_dreamer_amd/lib/libracecar_hip.so; cp $lib /tmp/ab_original.so
for r in 1 2; do for v in racing_dreamer_amd/lib/ab/*.so; do cp $v $lib; echo -n "$(basename $v .so): "; python tools/patch_store_bound.py 2>/dev/null | tail -1; done; done
cp /tmp/ab_original.so $lib
