// POD parameter blocks: the public definitions live in include/ryolo_params.h
#pragma once
#include "ryolo_params.h"
