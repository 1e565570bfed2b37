# Gram kernel durations of the product's three paths on the C4 table (2M x 64 fp64; GRAM_DTYPE=f32 for the float table):
# the plain pass (pbn_table_sse), the segmented pass of the score-data constructor (pbn_scoredata_create), the gathered segmented pass of
# a MutualInformation grouping (mi.hip ensure_full).  Usage (GPU box): bash tools/gram_paths.sh
echo "== plain (pbn_table_sse)"; bash tools/gram_timing.sh gram_plain | grep "gram_[a-z0-9_]*kernel\|per call"
echo "== segments (pbn_scoredata_create)"; GRAM_MODE=segments bash tools/gram_timing.sh gram_seg | grep "gram_[a-z0-9_]*kernel\|per call\|pbn_"
echo "== gather, 4 categories (MutualInformation grouping)"; GRAM_MODE=gather bash tools/gram_timing.sh gram_gather | grep "gram_[a-z0-9_]*kernel\|per call\|pbn_"
echo "== gather, 64 categories"; GRAM_MODE=gather GRAM_CARD=64 bash tools/gram_timing.sh gram_gather64 | grep "gram_[a-z0-9_]*kernel\|per call\|pbn_"
