// typedefs.h -- source-level drop-in for copra's include/typedefs.h:9-42: Eigen::Index and the trait that guards the
// perfect-forwarding constructors of the cost / constraint classes (true when every argument type is arithmetic).
#pragma once
#include "copra/EigenCompat.h"
#include <type_traits>

namespace copra {

template <typename... Ts>
struct is_all_arithmetic {
    static const bool value = std::conjunction<std::is_arithmetic<std::decay_t<Ts>>...>::value;
};

} // namespace copra
