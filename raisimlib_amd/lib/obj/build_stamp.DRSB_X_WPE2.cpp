extern "C" const char* rsb_source_hash(void) { return "244b83ef33506773cd88864a3507b4cc"; }
