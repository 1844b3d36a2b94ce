cd $GRAFT_REPO_ROOT
for L in "p2 3x3" "rpn 3x3" "sem 3x3 256->128" "sem 3x3 128->256" "p3 3x3" "res2 3x3" "p5 3x3"; do U2_BENCH_LAYERS="$L" tests/native/selftest bench2w 0 0x11000 0x21000 0x10000 0x21000 | grep LAYER | cut -c1-330; done
