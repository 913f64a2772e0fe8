"""Mirror of the reference's Backprojection_Loss tree (module names and signatures)."""
