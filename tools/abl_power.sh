for z in "" 1 "" 1; do for d in 0 5 6; do ABL_ZERO=$z ABL_ONE=$d python3 tools/ablate_conv_group.py 8 8 128 | tail -1 | sed "s/^/zero=[$z] /"; done; done
