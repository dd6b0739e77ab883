// shim_main.cpp -- `rcgpu-ffmpeg`: accepts the ffmpeg command line RAWcooked assembles (CLI/Output.cpp:81-332)
// and runs it on the MI355X encoder.  Use with an unmodified rawcooked:  rawcooked --bin-name /path/to/rcgpu-ffmpeg <dir>
// Exit status 0 = success, anything else is propagated by the reference (Output.cpp:356-374).
#include "rcgpu.h"
int main(int argc, char** argv) { return rcgpu_main_ffmpeg_argv(argc, argv); }
