/* see Rinternals.h in this directory: syntax-check stub only */
