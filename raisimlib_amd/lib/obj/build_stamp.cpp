extern "C" const char* rsb_source_hash(void) { return "a4f90b5ac8f95a192fe5559007d0ead1"; }
