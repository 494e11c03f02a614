MOD_sampler=ptdl
