// TEST INFRASTRUCTURE ONLY: libg2o_mini_types_sba.so -- what libg2o_types_sba.so is to g2o.  The out-of-line virtuals below are the
// key functions of the bundle-adjustment types: their vtables and typeinfo objects live HERE and nowhere else, so anything that
// names typeid(EdgeProjectXYZ2UV) (the adapter's device fast path) has to link this library, exactly as with the real g2o
// (the classes of /root/reference/g2o/types/sba/types_six_dof_expmap.h have their read / write in types_six_dof_expmap.cpp).
#include "g2o/types/sba/types_six_dof_expmap.h"
namespace g2o {
bool VertexSE3Expmap::write(std::ostream& os) const {
  const SE3Quat& T = estimate();
  for (int k = 0; k < 9; ++k) os << T.rotationMatrix().data()[k] << " ";
  for (int k = 0; k < 3; ++k) os << T.translation()[k] << " ";
  return os.good();
}
bool VertexSBAPointXYZ::write(std::ostream& os) const {
  os << _estimate[0] << " " << _estimate[1] << " " << _estimate[2];
  return os.good();
}
bool EdgeProjectXYZ2UV::write(std::ostream& os) const {
  os << _measurement[0] << " " << _measurement[1] << " ";
  for (int i = 0; i < 2; ++i)
    for (int j = i; j < 2; ++j) os << " " << _information(i, j);
  return os.good();
}
}  // namespace g2o
