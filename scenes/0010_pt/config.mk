MOD_sampler=pt
