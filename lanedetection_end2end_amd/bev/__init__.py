"""Mirror of the reference's Birds_Eye_View_Loss tree (module names and signatures)."""
