// TEST INFRASTRUCTURE ONLY: see ../functional.hpp
#pragma once
#include "../functional.hpp"
