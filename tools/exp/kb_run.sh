# timing of the one-instance-per-lane pass with parts of its memory traffic compiled out (tools/exp/lane_variants.sh build, with TAG / EXTRA)
TAG=kb1_ bash tools/exp/lane_variants.sh run "0 128 2 130 131"
