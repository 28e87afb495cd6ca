extern "C" const char* rsb_source_hash(void) { return "0d06ddd7d908272b6df6b2235c21849f"; }
