#pragma once
#include <cstddef>
#include "../../../eicos_amd/csrc/plans.hpp"
struct BasePlan { int *fsl, *bsl, *fidx16, *bidx16; int nfs, nfs_solo, nbs_solo, nbs, f_d16, b_d16; };
BasePlan make_base_plan(const eicos::TriPlan &pf, const eicos::TriPlan &pb, int N);
void free_base_plan(BasePlan &b);
size_t base_table_bytes(const eicos::TriPlan &pf, const eicos::TriPlan &pb);
void launch_base(int T, int grid, const BasePlan &bp, const double *UF, const double *UB, const double *invD, size_t sUF, size_t sUB, size_t sD,
                 const double *rhs, double *out, int N, int Npad, int nUF, int nUB, int reps);
