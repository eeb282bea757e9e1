#pragma once
#include <string>
#include "frayhip.h"
namespace frayhip_detail {
void set_error(const std::string& s);
}
