// flac_host.cpp -- placeholder translation unit; the FLAC host logic lives in flac_gpu.hip.
