# Generates g2o/config.h from the reference's own template (config.h.in) with
# cmake's configure_file -- the same tool/step the reference build uses
# (/root/reference/CMakeLists.txt: configure_file(config.h.in ...)).  No
# hand-written stand-in header is involved.  Options mirror a minimal static
# CSparse-only configuration.
set(G2O_HAVE_CSPARSE 1)
configure_file(${SRC} ${DST})
