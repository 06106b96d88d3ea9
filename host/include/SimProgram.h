// SimProgram.h — builds the reference's literal instruction stream for one operation (see host/src/SimProgram.cpp).
#ifndef HOMULATOR_SIMPROGRAM_H
#define HOMULATOR_SIMPROGRAM_H
#include "Addr.h"
#include "SimModel.h"

// op: HMULT | HROTATE | HADD | PMULT | PADD; label: the op's label ("test_hmult", ...: part of the rescale buffer names);
// `am`: the address plan the Operation constructor allocated; `inputs`: ct1.c0 / ct1.c1 / ct2.c0 / ct2.c1 / pt limb starts.
SimProgram buildSimProgram(const std::string &op, const std::string &label, uint32_t level, uint32_t alpha, Config *cfg,
                           const AddrManage &am, const std::map<std::string, std::vector<AddrType>> &inputs);
#endif
