#!/bin/bash
# Builds libu2seg_hip.so (gfx950 only) in-tree. Usage: build.sh [extra hipcc flags]   (U2_FORCE=1: recompile every source)
set -e
cd "$(dirname "$0")"
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -I. -I../../include -Wno-unused-result"
OBJS=""
PIDS=""
for f in conv_igemm conv_tile conv_halo conv_stream wgrad_halo wgrad_stream norm pool_resize losses roi select optim kmeans knn postprocess; do
  if [ -n "$U2_FORCE" ] || [ ! -f $f.o ] || [ $f.hip -nt $f.o ] || [ common.h -nt $f.o ] || [ conv_args.h -nt $f.o ] || [ ../../include/u2seg_hip.h -nt $f.o ]; then
    # compile to a temporary name so that a failed compile can never leave a stale object behind a "successful" link
    ( $HIPCC $FLAGS "$@" -c $f.hip -o $f.o.tmp && mv $f.o.tmp $f.o ) &
    PIDS="$PIDS $!"
  fi
  OBJS="$OBJS $f.o"
done
for p in $PIDS; do wait $p || { echo "build.sh: a HIP source failed to compile" >&2; exit 1; }; done
$HIPCC --offload-arch=gfx950 -shared -fPIC $OBJS -o libu2seg_hip.so
echo "built $(pwd)/libu2seg_hip.so ($(echo $PIDS | wc -w) of $(echo $OBJS | wc -w) sources compiled)"
