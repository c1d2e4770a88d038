from echoseal_amd.reliability import Q_Nmax  # noqa: F401
