# The gathered Gram of a MutualInformation grouping, 2M x 64 fp64 (GRAM_DTYPE=f32 for the float table), 4 and 64 configurations:
# the row-major mirror (default), the column gather (PBN_MI_MIRROR_MB=0) in its three launch orders (PBN_MI_GRAM_ORDER 0 =
# configuration-major, 1 = stripe-major, 2 = XCD-aligned stripe-major) and the older LDS-image kernel (PBN_GRAM_LDS=1).
# Usage (GPU box): bash tools/gram_orders.sh
for card in 4 64; do
  echo "== gather, $card categories, row-major mirror"; GRAM_MODE=gather GRAM_CARD=$card bash tools/gram_timing.sh gram_o | grep "gram_g[a-z0-9_]*kernel\|mirror_kernel\|pbn_mi"
  for o in 0 1 2; do
    echo "== gather, $card categories, columns, PBN_MI_GRAM_ORDER=$o"; PBN_MI_MIRROR_MB=0 PBN_MI_GRAM_ORDER=$o GRAM_MODE=gather GRAM_CARD=$card bash tools/gram_timing.sh gram_o | grep "gram_g[a-z0-9_]*kernel"
  done
  echo "== gather, $card categories, columns, LDS-image kernel (PBN_GRAM_LDS=1), order 2"; PBN_MI_MIRROR_MB=0 PBN_GRAM_LDS=1 GRAM_MODE=gather GRAM_CARD=$card bash tools/gram_timing.sh gram_o | grep "gram_[a-z0-9_]*kernel<"
done
