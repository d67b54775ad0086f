# timing of the roll-out's prefetch-depth variants (tools/exp/lane_variants.sh build, with TAG / EXTRA)
for kb in 1 4; do TAG=kb${kb}_ bash tools/exp/lane_variants.sh run "64 66"; done
