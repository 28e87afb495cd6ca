extern "C" const char* rsb_source_hash(void) { return "1acc282d51d359097b533cc7a3e3f449"; }
