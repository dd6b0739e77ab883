# Working on the integration patches (oracle/route_*.patch) without keeping any reference source in the repository.
#   bash tools/repatch.sh out   -> /tmp/rcp/a (the reference's files as they are) and /tmp/rcp/b (with the patches applied): edit b
#   bash tools/repatch.sh in    -> regenerates every oracle/route_*.patch from a and b (diff -U0: added lines only, no context)
set -e
REF=${REF:-/root/reference}; R=$(cd $(dirname $0)/.. && pwd); T=/tmp/rcp
declare -A FILES=( [route_b_output_cpp]=Source/CLI/Output.cpp [route_c_ffv1_frame_cpp]=Source/Lib/CoDec/FFV1/FFV1_Frame.cpp [route_c_matroska_cpp]=Source/Lib/Compressed/Matroska/Matroska.cpp
                   [route_c_filewriter_cpp]=Source/Lib/Utils/FileIO/FileWriter.cpp [route_c_track_cpp]=Source/Lib/Compressed/RAWcooked/Track.cpp [route_d_main_cpp]=Source/CLI/Main.cpp
                   [route_d_input_base_cpp]=Source/Lib/Utils/FileIO/Input_Base.cpp [route_d_dpx_cpp]=Source/Lib/Uncompressed/DPX/DPX.cpp
                   [linked_threadpool_h]=Source/Lib/ThirdParty/thread-pool/include/ThreadPool.h )
if [ "$1" = out ]; then
  rm -rf $T; mkdir -p $T/a $T/b
  for p in "${!FILES[@]}"; do f=${FILES[$p]}; mkdir -p $T/a/$(dirname $f) $T/b/$(dirname $f); cp $REF/$f $T/a/$f; cp $REF/$f $T/b/$f; [ -f $R/oracle/$p.patch ] && patch -s --binary -p1 -d $T/b < $R/oracle/$p.patch; done
  echo "edit the files under $T/b, then: bash tools/repatch.sh in"
elif [ "$1" = in ]; then
  for p in "${!FILES[@]}"; do f=${FILES[$p]}; (cd $T && diff -U0 --label a/$f --label b/$f a/$f b/$f > $R/oracle/$p.patch || true); n=$(grep -c '^-[^-]' $R/oracle/$p.patch || true); echo "$p: $(grep -c '^+[^+]' $R/oracle/$p.patch) added, $n removed"; done
else echo "usage: $0 out|in"; exit 1; fi
