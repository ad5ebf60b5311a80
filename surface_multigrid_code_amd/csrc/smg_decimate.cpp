// smg_decimate.cpp -- placeholder, replaced below in this round by the host decimator.
#include <string>
#include "smg_mesh.hpp"
namespace smg {
int decimate_level(const Mesh&, int, int, Mesh&, Csr&, std::string& err)
{
    err = "decimation-built levels are not available yet";
    return -1;
}
}  // namespace smg
