// ref_on_driver.cpp -- TEST INFRASTRUCTURE ONLY.
//
// Calls the reference's own CPU re-measurement of the O(N) ("O(1) radial") scalar path,
// SUN_CPU::SU::lattice_check_cpu (O(N)/suncl/suncpu.cpp:1805-1838), on a field read from a file.
// The reference's sources are compiled where they lie (see Makefile target `ref_on`); this file only
// fills in the public members those functions read -- no OpenCL device is touched: model's and
// GPU's constructors only zero their members (O(N)/suncl/suncl.cpp:82-106,
// O(N)/clinterface/clinterface.cpp:58-110).
//
//   qcdgpu_on_check n1 n2 n3 n4 precision z lambda zeta eta b skgroup cx cy cz ct field.bin
//
// field.bin holds n1*n2*n3*n4 values (float for precision 0, double for 1) in site order
// gid = y + z*n2 + t*n2*n3 + x*n2*n3*n4.  Prints: action per site, mean field, mean field^2,
// correlator 1 (step cx,cy,cz,ct), correlator 2 (diagonal), "%.17g" each.
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "suncl/suncl.h"
#include "suncl/suncpu.h"

int main(int argc, char** argv)
{
    if (argc != 17) {
        fprintf(stderr, "usage: %s n1 n2 n3 n4 precision z lambda zeta eta b skgroup cx cy cz ct field.bin\n", argv[0]);
        return 2;
    }
    model_CL::model* lat = new model_CL::model;
    model_CL::model::run_parameters* run = lat->run;
    for (int i = 0; i < 4; i++) run->lattice_domain_size[i] = run->lattice_full_size[i] = atoi(argv[1 + i]);
    run->precision = atoi(argv[5]) ? model_CL::model::model_precision_double : model_CL::model::model_precision_single;
    run->ON_z = atof(argv[6]);
    run->ON_lambda = atof(argv[7]);
    run->ON_zeta = atof(argv[8]);
    run->ON_eta = atof(argv[9]);
    run->ON_b = atof(argv[10]);
    run->ON_SK_group = (unsigned int)atoi(argv[11]);
    run->correlator_X = atoi(argv[12]);
    run->correlator_Y = atoi(argv[13]);
    run->correlator_Z = atoi(argv[14]);
    run->correlator_T = atoi(argv[15]);
    run->get_plaquettes_avr = true;
    run->get_correlators = true;
    run->lattice_type = 1;
    run->lattice_group = 1;
    // derived members, as model::model_lattice_init sets them (O(N)/suncl/suncl.cpp:1244,1264-1278)
    run->ON_sqrt_z_lambda_zeta = 0.25 * sqrt(run->ON_z / (run->ON_lambda * run->ON_zeta));
    const int n1 = run->lattice_domain_size[0], n2 = run->lattice_domain_size[1];
    const int n3 = run->lattice_domain_size[2], n4 = run->lattice_domain_size[3];
    lat->lattice_domain_n1 = n1;
    lat->lattice_domain_n2n3 = n2 * n3;
    lat->lattice_domain_n2n3n4 = n2 * n3 * n4;
    lat->lattice_full_site = (unsigned int)(n1 * n2 * n3 * n4);
    lat->lattice_domain_site = lat->lattice_full_site;
    lat->lattice_table_row_size = lat->lattice_full_site;

    const size_t sites = lat->lattice_full_site;
    const size_t bytes = sites * (run->precision == model_CL::model::model_precision_double ? 8 : 4);
    std::vector<unsigned int> table(bytes / 4);
    FILE* f = fopen(argv[16], "rb");
    if (!f || fread(table.data(), 1, bytes, f) != bytes) {
        fprintf(stderr, "cannot read %zu bytes from %s\n", bytes, argv[16]);
        return 1;
    }
    fclose(f);
    lat->lattice_pointer_last = table.data();
    for (int i = 0; i <= 16; i++) lat->Analysis[i].CPU_last_value = 0.0;

    SUN_CPU::SU* cpu = new SUN_CPU::SU;
    cpu->GPU0 = lat->GPU0;
    cpu->lattice_check_cpu(lat);

    printf("%.17g %.17g %.17g %.17g %.17g\n", lat->Analysis[DM_S_spat].CPU_last_value,
           lat->Analysis[DM_Plq_spat].CPU_last_value, lat->Analysis[DM_Plq_temp].CPU_last_value,
           lat->Analysis[DM_Correlator1].CPU_last_value, lat->Analysis[DM_Correlator2].CPU_last_value);
    fflush(stdout);
    _Exit(0);  // model's destructor writes a finish file and finalises an OpenCL device: skip it
}
