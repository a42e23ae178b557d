// ref_sun_driver.cpp -- TEST INFRASTRUCTURE ONLY.
//
// Calls the reference's own CPU re-measurement of an SU(3) or SU(2) configuration,
// SU3_CPU::SU::lattice_check_cpu (SU(N)/suncl/su3cpu.cpp:1616-1704) resp. SU2_CPU::SU::lattice_check_cpu
// (SU(N)/suncl/su2cpu.cpp:1120-1210): plaquettes, the F_mu_nu traces, the Wilson loop and the Polyakov loop, computed on the host from a link table in the reference's
// device layout.  The reference's sources are compiled where they lie (Makefile target `ref_sun`);
// this file only fills in the public members those functions read -- no OpenCL device is touched
// (model's and GPU's constructors only initialise members, SU(N)/suncl/suncl.cpp:142-214).
//
//   qcdgpu_sun_check group n1 n2 n3 n4 rowsize beta fmunu(0 none,1 H,2 E) index1 index2 wilsonR wilsonT table.bin [PHI OMEGA]
//
// PHI, OMEGA: the twisted boundary condition, which the re-measurement applies to every U_y it reads at x = n1 - 1
// (SU(N)/suncl/su3cpu.cpp:165-197, su2cpu.cpp:164-183).
//
// table.bin: 12 (SU(3)) or 4 (SU(2)) rows of rowsize double4 (SU(N)/suncl/su3_matrix_memory.cl:42-176,
// su2_matrix_memory.cl:55-142).  Prints one
// "name value" line per quantity, "%.17g".
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "suncl/suncl.h"
#include "suncl/su2cpu.h"
#include "suncl/su3cpu.h"

int main(int argc, char** argv)
{
    if (argc != 14 && argc != 16) {
        fprintf(stderr, "usage: %s group n1 n2 n3 n4 rowsize beta fmunu index1 index2 wilsonR wilsonT table.bin [PHI OMEGA]\n", argv[0]);
        return 2;
    }
    const double phi = argc == 16 ? atof(argv[14]) : 0.0, omega = argc == 16 ? atof(argv[15]) : 0.0;
    const int group = atoi(argv[1]);
    argv++;
    model_CL::model* lat = new model_CL::model;
    lat->lattice_nd = 4;
    lat->lattice_group = group;
    for (int i = 0; i < 4; i++) lat->lattice_domain_size[i] = lat->lattice_full_size[i] = atoi(argv[1 + i]);
    const int n1 = lat->lattice_full_size[0], n2 = lat->lattice_full_size[1], n3 = lat->lattice_full_size[2], n4 = lat->lattice_full_size[3];
    const unsigned int rowsize = (unsigned int)atoi(argv[5]);
    lat->BETA = atof(argv[6]);
    const int fm = atoi(argv[7]);
    lat->get_Fmunu = fm == 1;
    lat->get_F0mu = fm == 2;
    lat->Fmunu_index1 = atoi(argv[8]);
    lat->Fmunu_index2 = atoi(argv[9]);
    lat->wilson_R = atoi(argv[10]);
    lat->wilson_T = atoi(argv[11]);
    lat->get_wilson_loop = lat->wilson_R > 0;
    lat->get_plaquettes_avr = true;
    lat->precision = model_CL::model::model_precision_double;
    lat->PHI = phi;
    lat->OMEGA = omega;
    // derived members, as model::lattice_init / model_lattice_init set them (SU(N)/suncl/suncl.cpp:2212-2384)
    lat->lattice_domain_n1 = n1;
    lat->lattice_domain_n2n3 = n2 * n3;
    lat->lattice_domain_n2n3n4 = n2 * n3 * n4;
    lat->lattice_domain_n1n2n3 = n1 * n2 * n3;
    lat->lattice_full_n1n2n3 = n1 * n2 * n3;
    lat->lattice_full_site = (unsigned int)(n1 * n2 * n3 * n4);
    lat->lattice_domain_site = lat->lattice_full_site;
    lat->lattice_table_row_size = rowsize;
    lat->rowsize = rowsize;
    lat->rowsize4 = 4 * rowsize;

    const size_t words = (size_t)(group == 3 ? 12 : 4) * rowsize * 4 * 2;   // 32-bit words
    std::vector<unsigned int> table(words);
    FILE* f = fopen(argv[12], "rb");
    if (!f || fread(table.data(), 4, words, f) != words) {
        fprintf(stderr, "cannot read %zu bytes from %s\n", words * 4, argv[12]);
        return 1;
    }
    fclose(f);
    lat->lattice_pointer_last = table.data();

    if (group == 3) {
        SU3_CPU::SU* cpu = new SU3_CPU::SU;
        cpu->GPU0 = lat->GPU0;
        cpu->lattice_check_cpu(lat);
    } else {
        SU2_CPU::SU* cpu = new SU2_CPU::SU;
        cpu->GPU0 = lat->GPU0;
        cpu->lattice_check_cpu(lat);
    }

    static const struct { const char* name; int idx; } q[] = {
        {"S_spat", DM_S_spat}, {"S_temp", DM_S_temp}, {"Plq_spat", DM_Plq_spat}, {"Plq_temp", DM_Plq_temp},
        {"Wilson_loop", DM_Wilson_loop}, {"Polyakov_loop", DM_Polyakov_loop}, {"Polyakov_loop_im", DM_Polyakov_loop_im},
        {"Polyakov_loop_P2", DM_Polyakov_loop_P2}, {"Polyakov_loop_P4", DM_Polyakov_loop_P4},
        {"F_a_3_re", DM_Fmunu_xy_3_re}, {"F_a_3_im", DM_Fmunu_xy_3_im}, {"F_b_3_re", DM_Fmunu_xz_3_re}, {"F_b_3_im", DM_Fmunu_xz_3_im},
        {"F_c_3_re", DM_Fmunu_yz_3_re}, {"F_c_3_im", DM_Fmunu_yz_3_im}, {"F_a_8_re", DM_Fmunu_xy_8_re}, {"F_a_8_im", DM_Fmunu_xy_8_im},
        {"F_b_8_re", DM_Fmunu_xz_8_re}, {"F_b_8_im", DM_Fmunu_xz_8_im}, {"F_c_8_re", DM_Fmunu_yz_8_re}, {"F_c_8_im", DM_Fmunu_yz_8_im}};
    for (const auto& e : q) printf("%s %.17g\n", e.name, lat->Analysis[e.idx].CPU_last_value);
    fflush(stdout);
    _Exit(0);  // model's destructor finalises an OpenCL device: skip it
}
