// Basic.h — common types of the host layer (mirrors the role of the reference's include/Basic.h:1-23).
#ifndef HOMULATOR_BASIC_H
#define HOMULATOR_BASIC_H
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <ctime>
#include <fstream>
#include <iomanip>
#include <iostream>
#include <map>
#include <memory>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

// An address is a LINE number: one line = batchSize (256) coefficients, as in the reference
// (include/Basic.h:22).  In this build a line address is also a real location: see Arch.h.
typedef unsigned long long AddrType;

#define BASEADDRESS 0x00000001
#endif
