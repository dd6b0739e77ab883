// shim_main.cpp -- `rcgpu-ffmpeg`: accepts the ffmpeg command line RAWcooked assembles (CLI/Output.cpp:81-332)
// and runs it on the MI355X encoder.  Use with an unmodified rawcooked:  rawcooked --bin-name /path/to/rcgpu-ffmpeg <dir>
// Exit status 0 = success, anything else is propagated by the reference (Output.cpp:356-374).
//
// The process ends with the job: it leaves through _exit() once the output file is closed and the streams are flushed, so that the HIP
// runtime's own exit handlers do not run after everything has been given back already.  (Leaving the 140 GB of device memory and the
// pinned buffers to the kernel as well -- RCGPU_RELEASE_AT_EXIT=1 -- was measured and is slower: the kernel frees them after the file
// is closed, 1.8 s, while the job frees them beside the closing of the file.)
#include <cstdio>
#include <cstdlib>
#include <unistd.h>
#include "rcgpu.h"
int main(int argc, char** argv)
{
    const int rc = rcgpu_main_ffmpeg_argv(argc, argv);
    fflush(stdout); fflush(stderr);
    _exit(rc);
}
