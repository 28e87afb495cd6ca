extern "C" const char* rsb_source_hash(void) { return "15c729411753df1f5eaa63bc02ec2377"; }
