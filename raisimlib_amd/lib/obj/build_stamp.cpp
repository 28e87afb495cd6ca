extern "C" const char* rsb_source_hash(void) { return "5f8e7ae1e34bca57080fe68d1cd158b7"; }
