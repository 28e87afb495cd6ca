// rsb_internal.h — definitions shared by the host translation units of librsb.so (not installed).
#ifndef RSB_INTERNAL_H_
#define RSB_INTERNAL_H_

#include <string>

#include "rsb.h"

struct rsb_model {
  rsb_model_blob blob;
  int skipped_collisions;
};

namespace rsb {
void set_error(const std::string& s);
const char* last_error();
int validate_blob(const rsb_model_blob& b);
}  // namespace rsb

#endif
