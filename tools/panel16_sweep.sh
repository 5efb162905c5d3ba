for H in 512 1024; do for B in 256 500 512 777 1024; do
P16_ORACLE=1 P16_H=$H timeout 600 python tools/panel16_probe.py $B Uniform 2>&1 | grep "_ff._layers.0.bias\|lstm.bias_ih" | sed "s/^/H=$H B=$B /" | cut -c1-150
done; done
