# timing ablations of attn128x_kernel (tool build, results wrong): 1 no softmax, 2 no M stream, 4 no flush stores, 8 no Q fetch, 16 no tile DMA
for a in ${XP_ABLS:-0 1 2 4 8 16 12 28 29}; do
  echo "== abl $a"; M4D_LIB=abl M4D_ATTN_ABL=$a timeout 120 python tools/bench_attn.py cross_txt cross_i2v 2>&1 | grep cross
done
echo "== shipping"; timeout 120 python tools/bench_attn.py cross_txt cross_img cross_i2v 2>&1 | grep cross
echo "== shipping XP=0"; M4D_ATTN_XP=0 timeout 120 python tools/bench_attn.py cross_txt cross_img cross_i2v 2>&1 | grep cross
